"""What would MALL-resident weights be worth in the STEP?  The headline training step (1.5B, 16 x S = 256, hipGraph replay) with the 28
decoder layers sharing ONE layer's weight tensors (107 MB instead of 3 GB per pass: every GEMM after layer 0 finds its weights in the
256-MB Infinity Cache, the activations stay real and cold) against the real model, alternating in one process.  Timing only -- the
aliased model computes nonsense."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ps_slm_amd.streams import ensure_hw_queues
ensure_hw_queues()
import torch
from ps_slm_amd.config import DEFAULT_DS_CONFIG, ModelConfig, TrainConfig, load_ds_config
from ps_slm_amd.engine import TasuEngine
from ps_slm_amd.ps_slm import model_factory
from ps_slm_amd.synthetic import synthetic_text_batch

tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, gt_emb_noise=True, ctc_posterior=True, do_psd=True, use_fp16=True, batching_strategy="dynamic")
mc = ModelConfig(llm_path="synthetic:qwen2.5-1.5b", encoder_projector="linear-silu", encoder_dim=25055, llm_dim=1536)
model, _ = model_factory(tc, mc, device="cuda:0", init_seed=1234, keep_logits=False, with_encoder=False)
model.drop_prob = 0.0
core = model.core
core.use_graphs = True
engine = TasuEngine(model, load_ds_config(DEFAULT_DS_CONFIG))
engine.train()
raw = synthetic_text_batch(core.geo, 16, seed=1234, noise=False)
batch = dict(input_ids=raw["input_ids"], attention_mask=raw["attention_mask"], labels=raw["labels"], input_features=None, input_feature_length=None,
             GT=[" ".join(map(str, p)) for p in raw["post_ids"]])


def step():
    out, acc = engine(**batch)
    engine.backward(out.loss)
    engine.step()


def timed(n=20):
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


real_layers = list(core.llm.layers)
res = {"real": [], "layers_aliased": []}
for rnd in range(3):
    for name in ("real", "layers_aliased"):
        core.llm.layers[:] = real_layers if name == "real" else [real_layers[0]] * len(real_layers)
        core._graphs.clear(); core._graph_seen.clear()
        res[name].append(round(timed(), 3))
print(json.dumps({"ms_per_step": res}))
