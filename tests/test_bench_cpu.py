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
