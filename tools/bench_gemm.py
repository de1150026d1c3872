"""Micro-benchmark of tasu_gemm_nt_bf16_ws on the decoder's shapes (random data; HIP events on the launch stream).

  python tools/bench_gemm.py [M] [--cold] [--iters N] [--only name,name]

--cold rotates over enough distinct operand sets (>= 1.5 GB in total) that no launch finds its operands in the 256 MB
Infinity Cache -- the situation inside the training step, where every layer streams its own weights.
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps

ap = argparse.ArgumentParser()
ap.add_argument("M", nargs="?", type=int, default=4096)
ap.add_argument("--cold", action="store_true")
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--only", default="")
ap.add_argument("--vendor", action="store_true", help="also time torch.matmul (rocBLAS / hipBLASLt) on the same operand sets")
ap.add_argument("--check", action="store_true", help="compare against torch.matmul (rocBLAS) on the same bits")
ap.add_argument("--shape", action="append", default=[], help="extra M,N,K (repeatable); implies --only custom")
ap.add_argument("--pad", type=int, default=0, help="operand row pitch = K + pad elements (L2 channel experiment)")
args = ap.parse_args()
ops = HipOps()
M = args.M
shapes = [("qkv", M, 2048, 1536), ("o", M, 1536, 1536), ("gate_up", M, 17920, 1536), ("down", M, 1536, 8960),
          ("d_down", M, 8960, 1536), ("d_gate_up", M, 1536, 17920), ("d_qkv", M, 1536, 2048), ("lm_head", M, 151936, 1536),
          ("d_lm_head", M, 1536, 151936), ("proj1", 1664, 2048, 25088), ("wgrad1", 2048, 25088, 1664),
          ("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192)]
only = set(filter(None, args.only.split(",")))
if args.shape:
    shapes = [("custom", *map(int, sh.split(","))) for sh in args.shape]
    only = set()
res = []
for name, m, n, k in shapes:
    if only and name not in only:
        continue
    per_set = 2 * (m * k + n * k + m * n)
    nsets = max(2, min(16, -(-(3 << 29) // per_set))) if args.cold else 1
    ap0 = torch.randn(m, k + args.pad, device="cuda").to(torch.bfloat16)             # padded pitch; the operands are views
    bp0 = (torch.randn(n, k + args.pad, device="cuda") * k ** -0.5).to(torch.bfloat16)
    a0, b0 = ap0[:, :k], bp0[:, :k]
    sets = [(a0, b0, torch.empty(m, n, device="cuda", dtype=torch.bfloat16))]
    for _ in range(nsets - 1):
        sets.append((ap0.clone()[:, :k], bp0.clone()[:, :k], torch.empty(m, n, device="cuda", dtype=torch.bfloat16)))
    for i in range(max(3, nsets)):
        a, b, c = sets[i % nsets]
        ops.gemm(a, b, c, m, n, k)
    torch.cuda.synchronize()
    if args.check:
        ref = (a0.float() @ b0.float().t())
        err = float((sets[0][2].float() - ref).abs().max() / ref.abs().max())
        print(f"  check {name}: rel err {err:.2e}" + ("" if err < 1e-2 else "  <-- MISMATCH"), flush=True)
        del ref
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = args.iters
    e0.record()
    for i in range(iters):
        a, b, c = sets[i % nsets]
        ops.gemm(a, b, c, m, n, k)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * m * n * k / ms / 1e9
    extra = ""
    rec = dict(name=name, M=m, N=n, K=k, ms=round(ms, 4), tflops=round(tf, 1), sets=nsets)
    if args.vendor:
        for i in range(max(3, nsets)):
            a, b, c = sets[i % nsets]
            torch.matmul(a, b.t(), out=c)
        torch.cuda.synchronize()
        e0.record()
        for i in range(iters):
            a, b, c = sets[i % nsets]
            torch.matmul(a, b.t(), out=c)
        e1.record()
        torch.cuda.synchronize()
        vms = e0.elapsed_time(e1) / iters
        rec.update(vendor_ms=round(vms, 4), vendor_tflops=round(2.0 * m * n * k / vms / 1e9, 1))
        extra = f"   vendor {vms:8.3f} ms {rec['vendor_tflops']:7.1f} TF/s  ({vms / ms:.2f}x)"
    res.append(rec)
    print(f"{name:10s} M={m:5d} N={n:6d} K={k:6d}  {ms:8.3f} ms  {tf:7.1f} TF/s  ({nsets} sets){extra}", flush=True)
    del sets, a0, b0, a, b, c
    torch.cuda.empty_cache()
print(json.dumps(res))
