"""Kernel lab: the stream-K schedule of the 256 x 256 GEMM (tasu_gemm_nt_bf16_streamk) on the step's shapes whose tiles do not
fill whole rounds of CUs, against the tile policy without it: results (tolerance against the loader-wave kernel, bitwise
repeatable, flags left at zero), then both timed on cold rotating operand sets.

  python tools/lab_gemm_streamk.py [--iters N] [--only name,...]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps, GEMM_BF16, GEMM_F32, GEMM_RESID

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--only", default="")
args = ap.parse_args()
ops = HipOps()


def timeit(fn, sets, iters):
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


M = 4096
shapes = [("d_gate_up", M, 1536, 17920), ("down", M, 1536, 8960), ("d_down", M, 8960, 1536), ("gate_up_plain", M, 17920, 1536),
          ("lm_head", 2048, 151936, 1536), ("d_lm_head", 2048, 1536, 151936), ("o", M, 1536, 1536), ("qkv", M, 2048, 1536),
          ("edge", 1000, 1000, 1536), ("down7b", M, 3584, 18944), ("sq4096", 4096, 4096, 4096)]
only = set(filter(None, args.only.split(",")))
for name, m, n, k in shapes:
    if only and name not in only:
        continue
    per_set = 2 * (m * k + n * k + m * n)
    nsets = max(2, min(16, -(-(3 << 29) // per_set)))
    a0 = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    b0 = (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16)
    sets = [(a0, b0, torch.empty(m, n, device="cuda", dtype=torch.bfloat16))]
    for _ in range(nsets - 1):
        sets.append((a0.clone(), b0.clone(), torch.empty(m, n, device="cuda", dtype=torch.bfloat16)))
    ref = torch.zeros(m, n, device="cuda")
    ops.gemm_on("pipe128", a0, b0, ref, m, n, k, mode=GEMM_F32)
    out = []
    for mode in (GEMM_BF16, GEMM_F32, GEMM_RESID):
        dt = torch.bfloat16 if mode == GEMM_BF16 else torch.float32
        r = torch.randn(m, n, device="cuda") if mode == GEMM_RESID else None
        c1, c2 = torch.zeros(m, n, device="cuda", dtype=dt), torch.zeros(m, n, device="cuda", dtype=dt)
        ops.gemm_streamk(a0, b0, c1, m, n, k, resid=r, mode=mode)
        ops.gemm_streamk(a0, b0, c2, m, n, k, resid=r, mode=mode)
        torch.cuda.synchronize()
        want = ref if r is None else ref + r
        err = float((c1.float() - want).abs().max() / want.abs().max())
        out.append(f"{err:.1e}{'' if torch.equal(c1, c2) else ' NOT-REPEATABLE'}")
    flags = int(ops.gemm_ws[:4096 * 4].view(torch.int32).abs().sum())
    t_sk = timeit(lambda a, b, c: ops.gemm_streamk(a, b, c, m, n, k), sets, args.iters)
    t_pol = timeit(lambda a, b, c: ops.gemm(a, b, c, m, n, k), sets, args.iters)
    t_pp = timeit(lambda a, b, c: ops.gemm_on("pp256", a, b, c, m, n, k), sets, args.iters)
    t_192 = timeit(lambda a, b, c: ops.gemm_on("pipe192", a, b, c, m, n, k), sets, args.iters)
    fl = 2.0 * m * n * k
    print(f"{name:14s} {m}x{n}x{k}: streamk {t_sk:8.1f} us {fl / t_sk / 1e6:7.1f} TF | policy {t_pol:8.1f} us | pp256 whole tiles {t_pp:8.1f} | "
          f"pipe192 {t_192:8.1f} | rel err bf16/f32/resid {' '.join(out)} | flag words left: {flags}", flush=True)
