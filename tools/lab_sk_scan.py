"""Lab: per-K-tile cost and fixed cost of the stream-K schedule (96 tiles on 256 workgroups) against whole tiles (256 tiles)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps
ops = HipOps()


def timeit(fn, sets, iters=20):
    for i in range(max(3, len(sets))):
        fn(*sets[i % len(sets)])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(*sets[i % len(sets)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for m, n in ((4096, 1536), (4096, 4096), (4096, 3072), (2048, 1536)):
    for k in (2304, 4480, 8960, 17920, 35840):
        per_set = 2 * (m * k + n * k + m * n)
        nsets = max(2, min(8, -(-(3 << 29) // per_set)))
        sets = [(torch.randn(m, k, device="cuda").to(torch.bfloat16), (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16),
                 torch.empty(m, n, device="cuda", dtype=torch.bfloat16)) for _ in range(nsets)]
        t_sk = timeit(lambda a, b, c: ops.gemm_streamk(a, b, c, m, n, k), sets)
        t_pp = timeit(lambda a, b, c: ops.gemm_on("pp256", a, b, c, m, n, k), sets)
        t_192 = timeit(lambda a, b, c: ops.gemm_on("pipe192", a, b, c, m, n, k), sets)
        t_128 = timeit(lambda a, b, c: ops.gemm_on("pipe128", a, b, c, m, n, k), sets)
        tiles = (m // 256) * (n // 256)
        print(f"{m}x{n}x{k} ({tiles} tiles, {k // 64} K-tiles): streamk {t_sk:7.1f} us ({t_sk / (tiles * (k // 64) / 256):.2f} us per K-tile per WG) | "
              f"whole tiles {t_pp:7.1f} ({t_pp / (-(-tiles // 256) * (k // 64)):.2f}) | pipe192 {t_192:7.1f} | pipe128 {t_128:7.1f} | "
              f"TF streamk {2.0 * m * n * k / t_sk / 1e6:.0f}", flush=True)
        del sets
