"""The column remainders the tile policy runs as a second launch (d_down: 4096 x 768 x 1536 behind two whole rounds of 256 x 256
tiles) and a few whole shapes, on every named kernel: us per launch, cold rotating operands, graph replay."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps, GEMM_BF16
ops = HipOps()
bf = torch.bfloat16
L = 28
for name, (M, N, K) in {"d_down_rem": (4096, 768, 1536), "d_down_whole": (4096, 8960, 1536), "d_down_main": (4096, 8192, 1536),
                        "n1280": (4096, 1280, 1536), "n1024": (4096, 1024, 1536), "n512": (4096, 512, 1536)}.items():
    a = [torch.randn(M, K, device="cuda").to(bf) for _ in range(L)]
    w = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(bf) for _ in range(L)]
    c = torch.empty(M, N, device="cuda", dtype=bf)
    res = {}
    for kern in ("policy", "pp256", "pipe128", "pipe192", "pipe96"):
        def run():
            for l in range(L):
                if kern == "policy":
                    ops.gemm(a[l], w[l], c, M, N, K)
                else:
                    ops.gemm_on(kern, a[l], w[l], c, M, N, K)
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
    print(json.dumps({"shape": name, "MNK": (M, N, K), "us_tflops": res}), flush=True)
