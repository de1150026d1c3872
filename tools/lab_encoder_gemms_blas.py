"""The SANM encoder's four GEMM shapes through torch (hipBLASLt / rocBLAS) next to the library's policy: what a vendor kernel
gets on M = 16 x 504 rows with K = 512 / 2048 (bf16 out, bias; no residual epilogue on the vendor side).  Lab only."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps, GEMM_BF16
ops = HipOps()
bf = torch.bfloat16
L = 24
for M in (8064, 4032):
    for name, (N, K) in {"qkv": (1536, 512), "out": (512, 512), "w1": (2048, 512), "w2": (512, 2048)}.items():
        a = [torch.randn(M, K, device="cuda").to(bf) for _ in range(L)]
        w = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(bf) for _ in range(L)]
        bias = torch.randn(N, device="cuda").to(bf)
        c = torch.empty(M, N, device="cuda", dtype=bf)
        res = {}
        def ours():
            for l in range(L): ops.gemm(a[l], w[l], c, M, N, K, bias=bias, mode=GEMM_BF16)
        def blas():
            for l in range(L): torch.addmm(bias, a[l], w[l].t(), out=c)
        def blas_nobias():
            for l in range(L): torch.mm(a[l], w[l].t(), out=c)
        for kern, run in (("ours", ours), ("addmm", blas), ("mm", blas_nobias)):
            try:
                run(); torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g): run()
                g.replay(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): g.replay()
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 10 / L * 1e3
                res[kern] = (round(us, 1), round(2 * M * N * K / us / 1e6))
            except Exception as e:
                res[kern] = str(e)[:60]
        print(json.dumps({"M": M, "shape": name, "N": N, "K": K, "us_tflops": res}), flush=True)
