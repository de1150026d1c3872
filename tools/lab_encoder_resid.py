"""Encoder out / w2 projections (M = 16 x 504): fp32 residual epilogue against plain bf16 output on each named kernel. Lab only."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps, GEMM_BF16, GEMM_RESID
ops = HipOps()
bf = torch.bfloat16
M, L = 8064, 24
for name, (N, K) in {"out": (512, 512), "w2": (512, 2048)}.items():
    a = [torch.randn(M, K, device="cuda").to(bf) for _ in range(L)]
    w = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(bf) for _ in range(L)]
    bias = torch.randn(N, device="cuda").to(bf)
    r = torch.randn(M, N, device="cuda")
    for resid in (True, False):
        c = torch.empty(M, N, device="cuda", dtype=torch.float32 if resid else bf)
        res = {}
        for kern in ("policy", "pp256", "pipe128", "pipe192", "pipe96"):
            def run():
                for l in range(L):
                    kw = dict(bias=bias, resid=r if resid else None, mode=GEMM_RESID if resid else GEMM_BF16)
                    if kern == "policy": ops.gemm(a[l], w[l], c, M, N, K, **kw)
                    else: ops.gemm_on(kern, a[l], w[l], c, M, N, K, **kw)
            try:
                run(); torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g): run()
                g.replay(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): g.replay()
                e1.record(); torch.cuda.synchronize()
                res[kern] = round(e0.elapsed_time(e1) / 10 / L * 1e3, 1)
            except Exception as e:
                res[kern] = str(e)[:40]
        print(json.dumps({"shape": name, "resid": resid, "us": res}), flush=True)
