"""Micro-benchmark of the decode-step GEMMs (M = 64 rows) over 28 distinct weight sets (cold, like the layer loop).
TASU_SKINNY_BN / TASU_SKINNY_KS override the planner for sweeps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps

ops = HipOps()
M, L = 64, 28
ws = torch.zeros(32 * 64 * 151968, device="cuda")
shapes = [("qkv", 2048, 1536, False), ("o", 1536, 1536, False), ("gate_up+swiglu", 8960, 1536, True), ("down", 1536, 8960, False),
          ("lm_head", 151936, 1536, False)]
for name, n, k, sw in shapes:
    nl = 2 if name == "lm_head" else L
    rows = 2 * n if sw else n
    w = [(torch.randn(rows, k, device="cuda") * k ** -0.5).to(torch.bfloat16) for _ in range(nl)]
    a = torch.randn(M, k, device="cuda").to(torch.bfloat16)
    c = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
    def run():
        for i in range(nl):
            if sw:
                ops.gemm_skinny_swiglu(a, w[i], c, M, n, k, ws)
            else:
                ops.gemm_skinny(a, w[i], c, M, n, k, ws)
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps / nl * 1e3
    print(f"{name:16s} N={n:6d} K={k:5d}  {us:7.2f} us/launch-pair  {rows * k * 2 / us / 1e6:6.2f} TB/s", flush=True)
    del w
    torch.cuda.empty_cache()
