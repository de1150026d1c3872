"""The SANM encoder's four GEMM shapes (M = 16 x 504 rows, K = 512 or 2048) on every named kernel of tasu_gemm_nt_bf16_kernel
against the dispatcher's policy: us per launch over 70 rotating weight sets (cold operands, graph replay)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps, GEMM_BF16, GEMM_RESID
ops = HipOps()
bf = torch.bfloat16
M, L = 8064, 24
shapes = {"qkv": (1536, 512, False), "out": (512, 512, True), "w1": (2048, 512, False), "w2": (512, 2048, True)}
for name, (N, K, resid) in shapes.items():
    a = [torch.randn(M, K, device="cuda").to(bf) for _ in range(L)]
    w = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(bf) for _ in range(L)]
    bias = torch.randn(N, device="cuda").to(bf)
    r = torch.randn(M, N, device="cuda")
    c = torch.empty(M, N, device="cuda", dtype=torch.float32 if resid else bf)
    res = {}
    for kern in ("policy", "pp256", "pipe128", "pipe192", "pipe96"):
        def run():
            for l in range(L):
                if kern == "policy":
                    ops.gemm(a[l], w[l], c, M, N, K, bias=bias, resid=r if resid else None, mode=GEMM_RESID if resid else GEMM_BF16)
                else:
                    ops.gemm_on(kern, a[l], w[l], c, M, N, K, bias=bias, resid=r if resid else None, mode=GEMM_RESID if resid else GEMM_BF16)
        try:
            run(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run()
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 10 / L * 1e3
            res[kern] = (round(us, 1), round(2 * M * N * K / us / 1e6))
        except Exception as e:
            res[kern] = str(e)[:40]
    print(json.dumps({"shape": name, "N": N, "K": K, "us_tflops": res}), flush=True)
