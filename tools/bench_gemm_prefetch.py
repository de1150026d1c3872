"""Experiment: a chain of GEMMs on distinct cold operand sets, with the NEXT set's operands pulled through the
Infinity Cache by tasu_cache_prefetch on a side stream while the current GEMM runs.  Reports whole-chain time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps

ops = HipOps()
M = 4096
side = torch.cuda.Stream()
shapes = [("gate_up", M, 17920, 1536), ("down", M, 1536, 8960), ("d_down", M, 8960, 1536), ("d_gate_up", M, 1536, 17920),
          ("qkv", M, 2048, 1536), ("o", M, 1536, 1536)]
for name, m, n, k in shapes:
    per_set = 2 * (m * k + n * k + m * n)
    nsets = max(2, min(16, -(-(3 << 29) // per_set)))
    a0 = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    b0 = (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16)
    sets = [(a0.clone(), b0.clone(), torch.empty(m, n, device="cuda", dtype=torch.bfloat16)) for _ in range(nsets)]
    out = []
    for mode in ("none", "B/512/0", "B/512/1", "B/256/0", "B/1024/0", "AB/512/0"):
        iters = 48
        main = torch.cuda.current_stream()
        def run(n_it):
            for i in range(n_it):
                a, b, c = sets[i % nsets]
                if mode != "none":
                    what, blocks, pol = mode.split("/")
                    na, nb, _ = sets[(i + 1) % nsets]
                    side.wait_stream(main)            # fork: prefetch runs beside GEMM i
                    ops.cache_prefetch(nb, side, int(blocks), int(pol))
                    if what == "AB":
                        ops.cache_prefetch(na, side, int(blocks), int(pol))
                ops.gemm(a, b, c, m, n, k)
            main.wait_stream(side)
        run(nsets)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(iters)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        out.append(f"{mode} {ms*1e3:6.1f}us {2.0*m*n*k/ms/1e9:6.1f}")
    print(f"{name:10s} " + " | ".join(out), flush=True)
    del sets
    torch.cuda.empty_cache()
