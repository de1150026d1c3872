"""Kernel lab: the 256 x 256 ping-pong GEMM (csrc/gemm_pp.hip) against the loader-wave kernel (csrc/gemm_pipe.hip) on the
step's wide shapes: results compared on the same bits, then both timed on cold rotating operand sets.

  python tools/lab_gemm_pp.py [--iters N] [--only name,...] [--vendor]
"""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps, GEMM_BF16, GEMM_F32, GEMM_RESID

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--only", default="")
ap.add_argument("--vendor", action="store_true")
ap.add_argument("--warm", action="store_true", help="one operand set (replayed, cache-warm) instead of rotating cold sets")
args = ap.parse_args()
ops = HipOps()
st = lambda: torch.cuda.current_stream().cuda_stream


def pp(a, b, c, m, n, k, mode=GEMM_BF16, bias=None, resid=None):
    ops.gemm_on("pp256", a, b, c, m, n, k, bias=bias, resid=resid, mode=mode)


def pipe(a, b, c, m, n, k, mode=GEMM_BF16, bias=None, resid=None, bn=128):
    ops.gemm_on(f"pipe{bn}", a, b, c, m, n, k, bias=bias, resid=resid, mode=mode)


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
    return e0.elapsed_time(e1) / iters


M = 4096
shapes = [("d_down", M, 8960, 1536), ("gate_up_plain", M, 17920, 1536), ("n16384", M, 16384, 1536), ("qkv", M, 2048, 1536),
          ("lm_head", 2048, 151936, 1536), ("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192), ("edge", 1000, 1000, 1536),
          ("down", M, 1536, 8960)]
only = set(filter(None, args.only.split(",")))
res = []
for name, m, n, k in shapes:
    if only and name not in only:
        continue
    per_set = 2 * (m * k + n * k + m * n)
    nsets = 1 if args.warm else max(2, min(16, -(-(3 << 29) // per_set)))
    a0 = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    b0 = (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16)
    sets = [(a0, b0, torch.empty(m, n, device="cuda", dtype=torch.bfloat16))]
    for _ in range(nsets - 1):
        sets.append((a0.clone(), b0.clone(), torch.empty(m, n, device="cuda", dtype=torch.bfloat16)))
    # correctness on the same bits: bf16 output, fp32 output with bias, residual mode
    c_ref, c_new = torch.zeros(m, n, device="cuda", dtype=torch.bfloat16), torch.zeros(m, n, device="cuda", dtype=torch.bfloat16)
    bn = 192 if n == 1536 else 128
    pipe(a0, b0, c_ref, m, n, k, bn=bn)
    pp(a0, b0, c_new, m, n, k)
    torch.cuda.synchronize()
    bad = int((c_ref != c_new).sum())
    msg = f"bf16 mismatches {bad}"
    if m * n <= 4096 * 17920:
        bias = torch.randn(n, device="cuda").to(torch.bfloat16)
        f_ref, f_new = torch.zeros(m, n, device="cuda"), torch.zeros(m, n, device="cuda")
        pipe(a0, b0, f_ref, m, n, k, GEMM_F32, bias=bias, bn=bn)
        pp(a0, b0, f_new, m, n, k, GEMM_F32, bias=bias)
        r0 = torch.randn(m, n, device="cuda")
        r_ref, r_new = torch.zeros(m, n, device="cuda"), torch.zeros(m, n, device="cuda")
        pipe(a0, b0, r_ref, m, n, k, GEMM_RESID, resid=r0, bn=bn)
        pp(a0, b0, r_new, m, n, k, GEMM_RESID, resid=r0)
        torch.cuda.synchronize()
        msg += f", f32+bias {int((f_ref != f_new).sum())}, resid {int((r_ref != r_new).sum())}"
        del f_ref, f_new, r_ref, r_new, r0
    t_ref = timeit(lambda a, b, c: pipe(a, b, c, m, n, k, bn=bn), sets, args.iters)
    t_new = timeit(lambda a, b, c: pp(a, b, c, m, n, k), sets, args.iters)
    t_dis = timeit(lambda a, b, c: ops.gemm(a, b, c, m, n, k), sets, args.iters)
    fl = 2.0 * m * n * k / 1e9
    line = f"{name:14s} {m}x{n}x{k}: shipped {t_ref*1e3:8.1f} us {fl/t_ref:7.1f} TF | pp {t_new*1e3:8.1f} us {fl/t_new:7.1f} TF ({t_ref/t_new:.3f}x) | dispatcher {t_dis*1e3:8.1f} us | {msg}"
    if args.vendor:
        t_v = timeit(lambda a, b, c: torch.matmul(a, b.t(), out=c), sets, args.iters)
        line += f" | vendor {t_v*1e3:8.1f} us {fl/t_v:7.1f} TF"
    print(line, flush=True)
    res.append(dict(name=name, M=m, N=n, K=k, shipped_us=t_ref * 1e3, pp_us=t_new * 1e3, check=msg))
    del sets, a0, b0, c_ref, c_new
    torch.cuda.empty_cache()

# K-range slabs for the N = 1536 projections behind a long K
for name, m, n, k in (("down", M, 1536, 8960), ("d_gate_up", M, 1536, 17920)):
    if only and name not in only and "slabs" not in only:
        continue
    nsets = 1 if args.warm else 8
    sets = [(torch.randn(m, k, device="cuda").to(torch.bfloat16), (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16),
             torch.empty(m, n, device="cuda", dtype=torch.bfloat16)) for _ in range(nsets)]
    fl = 2.0 * m * n * k / 1e9
    a0, b0, c0 = sets[0]
    ref = torch.zeros(m, n, device="cuda")
    pipe(a0, b0, ref, m, n, k, GEMM_F32, bn=192)
    t_ref = timeit(lambda a, b, c: pipe(a, b, c, m, n, k, bn=192), sets, args.iters)
    line = f"{name:10s} {m}x{n}x{k}: pipe192 {t_ref*1e3:7.1f} us {fl/t_ref:7.1f} TF"
    for ks in (2, 4, 5, 7):
        if k % (128 * ks):
            continue
        ws = torch.zeros(ks, m, n, device="cuda")
        ops.gemm_slabs(a0, b0, ws, m, n, k, ks)
        torch.cuda.synchronize()
        err = float((ws.sum(0) - ref).abs().max() / ref.abs().max())
        t = timeit(lambda a, b, c: ops.gemm_slabs(a, b, ws, m, n, k, ks), sets, args.iters)
        line += f" | slabs x{ks} {t*1e3:7.1f} us {fl/t:7.1f} TF (err {err:.1e})"
        del ws
    print(line, flush=True)
    del sets

# gate|up with the SwiGLU epilogue: TASU_GEMM_GU_KERNEL=pipe / pp selects the kernel per process (run twice)
if not only or "gate_up" in only:
    m, I, k = M, 8960, 1536
    nsets = 1 if args.warm else 8
    mk = lambda: (torch.randn(m, k, device="cuda").to(torch.bfloat16), (torch.randn(2 * I, k, device="cuda") * k ** -0.5).to(torch.bfloat16),
                  torch.empty(m, 2 * I, device="cuda", dtype=torch.bfloat16), torch.empty(m, I, device="cuda", dtype=torch.bfloat16))
    sets = [mk() for _ in range(nsets)]
    t = timeit(lambda a, w, gu, act: ops.gemm_gate_up_swiglu(a, w, gu, act, m, I, k), sets, args.iters)
    fl = 2.0 * m * 2 * I * k / 1e9
    print(f"gate_up+swiglu {m}x{2*I}x{k} (TASU_GEMM_GU_KERNEL={os.environ.get('TASU_GEMM_GU_KERNEL', 'policy')}): {t*1e3:8.1f} us {fl/t:7.1f} TF", flush=True)
print(json.dumps(res))
