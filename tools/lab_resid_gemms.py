"""The decoder's two residual-add GEMMs (o projection 4096 x 1536 x 1536, down projection 4096 x 1536 x 8960; fp32 out = resid + bf16(acc))
and their plain-bf16 twins on every named kernel: us per launch over 28 rotating weight sets (cold operands, graph replay)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps, GEMM_BF16, GEMM_RESID
ops = HipOps()
bf = torch.bfloat16
M, L = 4096, 28
for name, (N, K) in {"o": (1536, 1536), "down": (1536, 8960)}.items():
    a = [torch.randn(M, K, device="cuda").to(bf) for _ in range(4)]
    w = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(bf) for _ in range(L)]
    r = torch.randn(M, N, device="cuda")
    for resid in (True, False):
        c = torch.empty(M, N, device="cuda", dtype=torch.float32 if resid else bf)
        res = {}
        for kern in ("policy", "pp256", "pipe128", "pipe192", "pipe96"):
            def run():
                for l in range(L):
                    kw = dict(resid=r if resid else None, mode=GEMM_RESID if resid else GEMM_BF16)
                    if kern == "policy": ops.gemm(a[l % 4], w[l], c, M, N, K, **kw)
                    else: ops.gemm_on(kern, a[l % 4], w[l], c, M, N, K, **kw)
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
                res[kern] = str(e)[:40]
        print(json.dumps({"shape": name, "resid": resid, "us_tflops": res}), flush=True)
