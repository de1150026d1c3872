"""Micro-benchmark of tasu_gemm_nt_bf16 on the decoder's shapes (random data; HIP events on the launch stream)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps

ops = HipOps()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
shapes = [("qkv", M, 2048, 1536), ("o", M, 1536, 1536), ("gate_up", M, 17920, 1536), ("down", M, 1536, 8960),
          ("d_down", M, 8960, 1536), ("d_gate_up", M, 1536, 17920), ("lm_head", M, 151936, 1536),
          ("d_lm_head", M, 1536, 151936), ("proj1", 1664, 2048, 25088), ("wgrad1", 2048, 25088, 1664),
          ("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192)]
res = []
for name, m, n, k in shapes:
    a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    b = (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16)
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, b, c, m, n, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 20
    e0.record()
    for _ in range(iters):
        ops.gemm(a, b, c, m, n, k)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * m * n * k / ms / 1e9
    res.append(dict(name=name, M=m, N=n, K=k, ms=round(ms, 4), tflops=round(tf, 1)))
    print(f"{name:10s} M={m:5d} N={n:6d} K={k:6d}  {ms:8.3f} ms  {tf:7.1f} TF/s", flush=True)
    del a, b, c
print(json.dumps(res))
