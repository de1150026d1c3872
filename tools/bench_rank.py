"""Microbenchmark of tasu_gemm_nt_rank at the LoRA shapes (events around N back-to-back launches; cold = rotating operands)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps

ops = HipOps()
shapes = [(4096, 64, 1536, 0, 0), (4096, 64, 8960, 0, 0), (4096, 64, 256, 0, 0), (1536, 64, 4096, 1, 0), (256, 64, 4096, 1, 0),
          (8960, 64, 4096, 1, 0), (1536, 64, 4096, 1, 1), (8960, 64, 4096, 1, 1)]
for M, N, K, f32, tr in shapes:
    nbuf = 8
    a = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(nbuf)]
    b = [torch.randn(N, K, device="cuda").bfloat16() for _ in range(nbuf)]
    c = torch.empty((N, M) if tr else (M, N), dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
    for mode in ("warm", "cold"):
        for _ in range(3):
            ops.gemm_rank(a[0], b[0], c, M, N, K, f32, tr)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 40
        e0.record()
        for i in range(reps):
            j = i % nbuf if mode == "cold" else 0
            ops.gemm_rank(a[j], b[j], c, M, N, K, f32, tr)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"M={M} N={N} K={K} f32={f32} tr={tr} {mode}: {us:.1f} us  A-stream {M*K*2/us/1e6:.2f} TB/s")
    # same through the tile GEMM for comparison
    c2 = torch.empty(M, N, dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
    for _ in range(3):
        ops.gemm(a[0], b[0], c2, M, N, K, mode=1 if f32 else 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(40):
        ops.gemm(a[i % nbuf], b[i % nbuf], c2, M, N, K, mode=1 if f32 else 0)
    e1.record()
    torch.cuda.synchronize()
    print(f"   tile GEMM cold: {e0.elapsed_time(e1) * 1e3 / 40:.1f} us")
