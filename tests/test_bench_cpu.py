"""bench.py's N > 1 bookkeeping without a GPU: ``train_leg`` driven by two gloo ranks on the CPU operator double at the "mid"
geometry (barriers around the timed region, MAX-reduce of the wall time and of the exposed all-reduce time, the record on rank 0
only, whole-job value = world x per-rank batch), and the launch check of ``--gpus N`` against WORLD_SIZE.  The first real 8-GPU
run of the driver exercises exactly this code with nccl (= RCCL) instead of gloo and HipOps instead of the double."""
import os
import sys
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from conftest import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from fake_ops import FakeOps
        args = SimpleNamespace(drop_prob=0.0, no_graphs=True)
        rec = bench.train_leg(args, "mid", "text", 2, 2, 1, world, rank, 0, False, device="cpu", ops=FakeOps())
        lrec = bench.train_leg(args, "mid", "text", 2, 2, 1, world, rank, 0, False, device="cpu", ops=FakeOps(), lora=True)   # --lora at N > 1
        arec = bench.train_leg(args, "mid", "audio", 2, 2, 1, world, rank, 0, False, device="cpu", ops=FakeOps())              # --path audio at N > 1
        ret[rank] = rec
        ret[10 + rank] = lrec
        ret[20 + rank] = arec
    finally:
        dist.destroy_process_group()


def test_train_leg_two_ranks_over_gloo():
    world, port = 2, free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    assert r1 is None                                              # only rank 0 builds the record
    assert r0["unit"] == "utterances/s" and r0["config"]["parallelism"] == "dp2" and r0["config"]["per_gpu_batch"] == 2
    # whole-job value: world x per-rank batch x steps / MAX-over-ranks wall time
    assert r0["value"] == pytest.approx(2 * 2 * 2 / (r0["ms_per_step"] * 2 * 1e-3), rel=2e-3)
    assert r0["allreduce_exposed_ms"] >= 0.0 and "Expected on 8 xGMI-connected GPUs" in r0["allreduce_note"]
    assert r0["roofline"]["bound"] == "mfma" and r0["config"]["seq_len"] == 256
    assert 0.0 < r0["config"]["final_loss"] < 20.0
    # the use_peft recipe through the same bookkeeping: the adapters' ranges ride in the exchange, FLOPs include the rank columns
    l0 = ret[10]
    assert ret[11] is None and "LoRA recipe" in l0["config"]["workload"] and l0["config"]["parallelism"] == "dp2"
    assert l0["allreduce_exposed_ms"] >= 0.0 and 0.0 < l0["config"]["final_loss"] < 20.0
    assert "bucket exchanged in" in l0["allreduce_note"]
    # config 4's path (audio-SFT: encoder -> PSD -> projector -> LLM) through the same bookkeeping
    a0 = ret[20]
    assert ret[21] is None and "audio-SFT step" in a0["config"]["workload"] and a0["config"]["parallelism"] == "dp2"
    assert a0["allreduce_exposed_ms"] >= 0.0 and 0.0 < a0["config"]["final_loss"] < 20.0


def test_launch_check_refuses_a_mismatched_world():
    import bench
    bench.check_launch(1, 1)
    bench.check_launch(8, 8)
    with pytest.raises(SystemExit, match="torch.distributed.run"):
        bench.check_launch(1, 8)
    with pytest.raises(SystemExit, match="WORLD_SIZE is 4"):
        bench.check_launch(4, 8)
    with pytest.raises(SystemExit):
        bench.check_launch(2, 1)


def _stub_full_record():
    """A full record of the shape main() builds, with prose fields far longer than the real ones."""
    import bench
    prose = "x" * 3000
    roof = {"bound": "mfma", "achieved": 1061.0, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.4244, "traffic": 291864737,
            "traffic_source": prose, "kernel": prose, "launches_per_step": 235, "avg_launch_us": 94.59, "gemm_ms_per_step": 22.2,
            "launch": "hipGraph replay", "passes_note": prose, "whole_step_tflops": 900.0, "whole_step_frac": 0.34,
            "whole_step_frac_at_survey_flops": 0.373, "flops_note": prose}
    sub = {"value": 1.0, "unit": "utterances/s", "ms_per_step": 2.0, "config": {"workload": prose}, "roofline": dict(roof),
           "side_streams": prose, "allreduce_note": prose}
    full = {"metric": "train utterances/sec (Qwen2.5-1.5B align)", "value": 570.88, "unit": "utterances/s", "n_gpus": 1, "steps": 20,
            "warmup": 3, "ms_per_step": 28.027, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic", "config": {"workload": prose, "per_gpu_batch": 16, "seq_len": 256, "parallelism": "dp1", "final_loss": 9.4},
            "roofline": roof, "decode": dict(sub, unit="tokens/s", roofline={"bound": "hbm", "frac": 0.276, "note": prose}),
            "cpu_baseline": {"value": 0.22, "unit": "utterances/s", "cores": 64, "kind": "port", "sample": prose},
            "cpu_baselines": {k: {"value": 0.4, "unit": "tokens/s", "cores": 64, "kind": "port", "sample": prose}
                              for k in ("train_B16", "decode_B1", "decode_B16")},
            "data_path": {"text_only_utterances_per_s": 558.0, "audio_wav_utterances_per_s": 237.5, "note": prose},
            "wall_seconds": {"headline+decode": 2.5, "cpu_baseline": 20.7}}
    for name in bench.SUB_METRICS:
        full[name] = dict(sub, decode=dict(sub)) if name == "qwen2.5-7b" else dict(sub)
    full["exchange_1rank"].update(collective={"ranks": 1, "library": prose}, allreduce_exposed_ms=0.07)
    return full


def _check_line(text):
    import json
    assert "\n" not in text.rstrip("\n") and len(text.encode()) < 8000
    line = json.loads(text, parse_constant=lambda c: pytest.fail(f"non-strict JSON constant {c}"))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in line, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert set(line["config"]) >= {"workload", "per_gpu_batch", "seq_len", "parallelism"}

    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(s) for s in strings(line)) <= 120          # (the driver truncates longer strings)
    return line


def test_stdout_line_stays_under_the_drivers_limit(tmp_path, capfd):
    """the driver keeps ~8,000 characters of stdout and parses the line from them.  Whatever the sub-records
    hold, ONE compact line < 8,000 bytes goes to stdout, strict JSON, with roofline and cpu_baseline at top level; the complete
    record goes to stderr and to a file."""
    import json

    import bench
    full = _stub_full_record()
    assert len(json.dumps(full)) > 50_000
    r, w = os.pipe()
    bench.emit(full, w, os.path.relpath(str(tmp_path / "full.json"), bench.ROOT))
    os.close(w)
    text = os.read(r, 1 << 20).decode()
    os.close(r)
    line = _check_line(text)
    assert line["value"] == 570.88 and line["roofline"]["frac"] == 0.4244 and line["roofline"]["launches_per_step"] == 235
    assert line["cpu_baseline"] == {"value": 0.22, "unit": "utterances/s", "cores": 64, "kind": "port", "sample": "x" * 117 + "..."}
    assert line["digest"]["decode"]["roofline_frac"] == 0.276 and line["digest"]["qwen2.5-7b_decode"]["value"] == 1.0
    assert line["digest"]["exchange_1rank"]["allreduce_exposed_ms"] == 0.07
    assert json.load(open(tmp_path / "full.json")) == full      # nothing is lost: the whole record is on disk ...
    assert "FULL_RECORD " in capfd.readouterr().err              # ... and on stderr


def test_stdout_line_of_a_multi_rank_record():
    """The N > 1 line (train_leg's record with the exchange fields) obeys the same limit: collective.ranks and
    allreduce_exposed_ms at top level."""
    import json

    import bench
    full = _stub_full_record()
    for name in bench.SUB_METRICS:
        full.pop(name)
    for k in ("cpu_baseline", "cpu_baselines", "data_path", "decode"):
        full.pop(k)
    full.update(n_gpus=8, collective={"ranks": 8, "library": "RCCL 2.26.6", "note": "y" * 2000}, allreduce_exposed_ms=0.41,
                allreduce_note="z" * 2000)
    full["config"]["parallelism"] = "dp8"
    line = _check_line(json.dumps(bench.compact_line(full)))
    assert line["n_gpus"] == 8 and line["collective"]["ranks"] == 8 and line["allreduce_exposed_ms"] == 0.41
    assert line["config"]["parallelism"] == "dp8" and "cpu_baseline" not in line
