"""Micro-benchmark of the decode step's per-layer ops at Qwen2.5-1.5B geometry (M = 64 beam rows) over 28 distinct weight sets
(cold, like the layer loop), hipGraph-replayed: the split-K + finish kernels (csrc/gemm_skinny.hip) against the single-launch
streaming kernels (csrc/gemm_stream.hip).  Usage: python tools/bench_decode_layer.py [frag|stream|skinny|both]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps

which = sys.argv[1] if len(sys.argv) > 1 else "both"
ops = HipOps()
M, L, D, I, H, G, V, ctx = 64, 28, 1536, 8960, 12, 2, 151936, 328
bf, f32 = torch.bfloat16, torch.float32
ws = torch.zeros(32 * 64 * 151968, device="cuda")
rn = lambda *s, k=1.0: (torch.randn(*s, device="cuda") * k).to(bf)
LD, W = (H + 2 * G) * 128, G * 128
wqkv = [rn(LD, D, k=D ** -0.5) for _ in range(L)]
bq = rn(LD)
wo = [rn(D, D, k=D ** -0.5) for _ in range(L)]
wgu = [rn(2 * I, D, k=D ** -0.5) for _ in range(L)]
wd = [rn(D, I, k=I ** -0.5) for _ in range(L)]
head = rn(V, D, k=D ** -0.5)
xn, ao, act = rn(M, D), rn(M, D), rn(M, I)
x, x2 = torch.randn(M, D, device="cuda"), torch.randn(M, D, device="cuda")
qkv = torch.zeros(M, LD, device="cuda", dtype=bf)
ang = torch.randn(M, 64, device="cuda")
cos, sin = torch.cos(ang), torch.sin(ang)
kc, vc = torch.zeros(M * ctx * W, device="cuda", dtype=bf), torch.zeros(M * ctx * W, device="cuda", dtype=bf)
pos = torch.full((M,), 200, device="cuda", dtype=torch.int32)
nw = torch.ones(D, device="cuda")
logits = torch.zeros(M, V, device="cuda", dtype=bf)
y = torch.zeros(M, D, device="cuda", dtype=bf)


def timed(name, fn, n, nbytes):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps / n * 1e3
    print(f"  {name:34s} {us:7.2f} us   {nbytes / us / 1e6:5.2f} TB/s of weights", flush=True)
    return us


for mode in (["frag", "stream", "skinny"] if which == "both" else [which]):
    ops.use_stream = mode != "skinny"
    if mode == "frag":                 # weights re-laid out in MFMA order + activations travelling in it
        for i in range(L):
            ops.register_decode_weight(wqkv[i], "qkv", LD, H, G)
            ops.register_decode_weight(wo[i], "plain", D)
            ops.register_decode_weight(wgu[i], "swiglu", I)
            ops.register_decode_weight(wd[i], "plain", D, slabs_ok=True)
        ops.register_decode_weight(head, "plain", V)
        assert ops.begin_decode(D, D, I)
    else:
        ops.end_decode()
        ops._frag = {}
    print(mode)
    tot = 0.0
    tot += timed("qkv + bias + rope + append", lambda: [ops.gemm_skinny_qkv_rope(xn, wqkv[i], bq, qkv, M, H, G, D, cos, sin, kc, vc, pos, ctx, ws) for i in range(L)], L, LD * D * 2)
    tot += timed("o + residual + norm", lambda: [ops.gemm_skinny_norm(ao, wo[i], x2, x, M, D, D, nw, y, 1e-6, ws) for i in range(L)], L, D * D * 2)
    tot += timed("gate|up + swiglu", lambda: [ops.gemm_skinny_swiglu(xn, wgu[i], act, M, I, D, ws) for i in range(L)], L, 2 * I * D * 2)
    tot += timed("down + residual + norm", lambda: [ops.gemm_skinny_norm(act, wd[i], x, x2, M, D, I, nw, y, 1e-6, ws) for i in range(L)], L, D * I * 2)
    print(f"  layer GEMMs total {tot:.1f} us -> x28 = {tot * 28 / 1e3:.2f} ms per position")
    timed("lm_head", lambda: ops.gemm_skinny(xn, head, logits, M, V, D, ws), 1, V * D * 2)
