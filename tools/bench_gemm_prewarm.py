"""Experiment: does a GEMM on cold operands reach its warm rate when the weight operand (or both) has just been pulled
through the Infinity Cache by a streaming read?  Times ONLY the GEMM launches (event pair per launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps

ops = HipOps()
M = 4096
shapes = [("gate_up", M, 17920, 1536), ("down", M, 1536, 8960), ("d_down", M, 8960, 1536), ("d_gate_up", M, 1536, 17920),
          ("qkv", M, 2048, 1536), ("o", M, 1536, 1536)]
for name, m, n, k in shapes:
    per_set = 2 * (m * k + n * k + m * n)
    nsets = max(2, min(16, -(-(3 << 29) // per_set)))
    a0 = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    b0 = (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16)
    sets = [(a0.clone(), b0.clone(), torch.empty(m, n, device="cuda", dtype=torch.bfloat16)) for _ in range(nsets)]
    out = []
    for mode in ("cold", "warmB", "writeA", "writeA+warmB", "warmAB", "same"):
        tot = 0.0
        iters = 40
        evs = []
        for i in range(iters + 4):
            a, b, c = sets[0] if mode == "same" else sets[i % nsets]
            if mode.startswith("writeA"):
                a.copy_(sets[(i + 3) % nsets][0])          # A freshly WRITTEN by the previous kernel (source is another set)
            if mode in ("warmB", "warmAB", "writeA+warmB"):
                b.view(torch.int32).max()
            if mode == "warmAB":
                a.view(torch.int32).max()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm(a, b, c, m, n, k)
            e1.record()
            if i >= 4:
                evs.append((e0, e1))
        torch.cuda.synchronize()
        ms = sum(x.elapsed_time(y) for x, y in evs) / len(evs)
        out.append(f"{mode} {ms*1e3:7.1f} us {2.0*m*n*k/ms/1e9:7.1f} TF/s")
    print(f"{name:10s} " + " | ".join(out), flush=True)
    del sets
    torch.cuda.empty_cache()
