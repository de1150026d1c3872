import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps
ops = HipOps()
def timeit(fn, sets, iters=20):
    for i in range(max(3, len(sets))): fn(*sets[i % len(sets)])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fn(*sets[i % len(sets)])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for m in (1024, 2048, 3072, 3584, 4096):
    for n, k in ((1536, 8960), (1536, 17920)):
        per_set = 2 * (m * k + n * k + m * n)
        nsets = max(2, min(8, -(-(3 << 29) // per_set)))
        sets = [(torch.randn(m, k, device="cuda").to(torch.bfloat16), (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16),
                 torch.empty(m, n, device="cuda", dtype=torch.bfloat16)) for _ in range(nsets)]
        t = {}
        t["sk"] = timeit(lambda a, b, c: ops.gemm_streamk(a, b, c, m, n, k), sets)
        t["pp"] = timeit(lambda a, b, c: ops.gemm_on("pp256", a, b, c, m, n, k), sets)
        t["192"] = timeit(lambda a, b, c: ops.gemm_on("pipe192", a, b, c, m, n, k), sets)
        t["128"] = timeit(lambda a, b, c: ops.gemm_on("pipe128", a, b, c, m, n, k), sets)
        t["pol"] = timeit(lambda a, b, c: ops.gemm(a, b, c, m, n, k), sets)
        print(f"{m}x{n}x{k}: streamk {t['sk']:7.1f} | whole 256x256 {t['pp']:7.1f} | pipe192 {t['192']:7.1f} | pipe128 {t['128']:7.1f} | policy {t['pol']:7.1f}", flush=True)
