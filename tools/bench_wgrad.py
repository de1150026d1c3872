"""The projector's Linear1 weight gradient GEMM (fp32 output straight into the bucket): dW1[2048, 25088] = dh1^T[2048, 1664] . xn^T[25088, 1664]^T,
and dW2[1536, 2048]; the dispatcher's choice against named kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps, GEMM_F32
ops = HipOps()
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for (M, N, K) in ((2048, 25088, 1664), (1536, 2048, 1664), (2048, 25088, 1024)):
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    c = torch.empty(M, N, dtype=torch.float32, device="cuda")
    us = bench(lambda: ops.gemm(a, b, c, M, N, K, mode=GEMM_F32))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: dispatcher {us:.1f} us = {fl / us / 1e6:.0f} TFLOP/s; C write alone = {M * N * 4 / 1e6:.0f} MB -> {M * N * 4 / us / 1e6:.2f} TB/s")
    for kname in ("pp256", "pipe128", "pipe192", "pipe96"):
        try:
            us = bench(lambda: ops.gemm_on(kname, a, b, c, M, N, K, mode=GEMM_F32))
            print(f"    {kname}: {us:.1f} us = {fl / us / 1e6:.0f} TFLOP/s")
        except Exception as e:
            print(f"    {kname}: {e}")
    cb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    us = bench(lambda: ops.gemm(a, b, cb, M, N, K))
    print(f"    bf16 output (dispatcher): {us:.1f} us")
