"""us per launch of the adapters' weight-gradient GEMMs at the step's shapes: tasu_gemm_tn_rank on the row-major operand against
tasu_gemm_nt_rank on a transposed copy (+ the transpose that path needs).  Graph replay over rotating operands."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps
ops = HipOps()
bf = torch.bfloat16
K, R, L = 4096, 64, 8
for M in (256, 1536, 8960):
    at = [torch.randn(K, M, device="cuda").to(bf) for _ in range(L)]
    a_t = [x.t().contiguous() for x in at]
    b = torch.randn(R, K, device="cuda").to(bf)
    c = torch.empty(M, R, device="cuda")
    tmp = torch.empty(M, K, device="cuda", dtype=bf)
    res = {}
    for name, fn in (("tn", lambda l: ops.gemm_rank_tn(at[l], b, c, M, R, K)),
                     ("tn_tstore", lambda l: ops.gemm_rank_tn(at[l], b, c.view(R, M), M, R, K, transposed=True)),
                     ("nt", lambda l: ops.gemm_rank(a_t[l], b, c, M, R, K, f32=True)),
                     ("transpose", lambda l: ops.transpose(at[l], tmp, K, M, K, M))):
        for l in range(L): fn(l)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for l in range(L): fn(l)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g.replay()
        e1.record(); torch.cuda.synchronize()
        res[name] = round(e0.elapsed_time(e1) / 10 / L * 1e3, 1)
    print(json.dumps({"M_out": M, "K": K, "us": res}), flush=True)
