"""Is the graph-replayed LoRA step's time a property of the step or of what the process did before?  (round 6: 44.5 ms inside the
full bench run, 77 ms as the first leg of a process, 44.7 ms with eager launches)
    python tools/lab_lora_graph_order.py first|after_text|after_decode|after_variable|after_audio|after_all|eager|exchange_first"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ps_slm_amd import streams  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "first"
streams.ensure_hw_queues()
args = argparse.Namespace(steps=10, warmup=2, batch=16, model="qwen2.5-1.5b", path="text", no_cpu_baseline=True, drop_prob=0.0,
                          no_graphs=mode in ("eager", "after_all_eager_lora"), no_decode=True, blank_biased=False, blank_bias=13.0, no_encoder_ahead=False, lora=True,
                          no_extra=True, no_data_path=True, gpus=1)
import torch  # noqa: E402
torch.cuda.set_device(0)
out = {}
if mode == "after_text":
    r = bench.train_leg(args, "qwen2.5-1.5b", "text", 16, 6, 2, 1, 0, 0, False)
    out["text_ms"] = r["ms_per_step"]
if mode == "exchange_first":                       # the N > 1 gradient-exchange stream (1-rank RCCL) as the first workload of a process
    args.lora = False
    r = bench.train_leg(args, "qwen2.5-1.5b", "text", 16, 10, 3, 1, 0, 0, False, force_exchange=True)
    print(json.dumps({"mode": mode, "exchange_ms": r["ms_per_step"], "exposed_ms": r.get("allreduce_exposed_ms"), "side_streams": r.get("side_streams")}))
    sys.exit(0)
if mode == "after_decode":
    r = bench.train_leg(args, "qwen2.5-1.5b", "text", 16, 6, 2, 1, 0, 0, True)
    out["text_ms"] = r["ms_per_step"]
if mode == "after_variable":
    r = bench.train_leg(args, "qwen2.5-1.5b", "text", 16, 32, 24, 1, 0, 0, False, variable=True)
    out["variable_ms"] = r["ms_per_step"]
if mode in ("after_all", "after_all_eager_lora"):
    args.no_graphs = False
    bench.train_leg(args, "qwen2.5-1.5b", "text", 16, 6, 2, 1, 0, 0, True)
    bench.train_leg(args, "qwen2.5-1.5b", "text", 16, 32, 24, 1, 0, 0, False, variable=True)
    bench.train_leg(args, "qwen2.5-1.5b", "audio", 16, 5, 2, 1, 0, 0, False)
    bench.train_leg(args, "qwen2.5-1.5b", "audio", 16, 5, 2, 1, 0, 0, False, blank_biased=True)
if mode == "after_audio":
    r = bench.train_leg(args, "qwen2.5-1.5b", "audio", 16, 5, 2, 1, 0, 0, False, blank_biased=True)
    out["audio_ms"] = r["ms_per_step"]
if os.environ.get("LAB_GC") == "1":
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    out["mem_gb_before_lora"] = round(torch.cuda.memory_allocated() / 2 ** 30, 1)
else:
    out["mem_gb_before_lora"] = round(torch.cuda.memory_allocated() / 2 ** 30, 1)
args.no_graphs = mode in ("eager", "after_all_eager_lora")
r = bench.train_leg(args, "qwen2.5-1.5b", "text", 16, 10, 2, 1, 0, 0, False, lora=True)
out.update(mode=mode, lora_ms=r["ms_per_step"], side_streams=r.get("side_streams"))
print(json.dumps(out))
