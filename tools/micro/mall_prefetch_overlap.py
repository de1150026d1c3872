"""Can the decode layer hide its MLP weight streaming under its latency-bound kernels?  Per layer, after the q|k|v projection a
side stream reads the layer's gate|up and down weights (a plain reduction kernel: the reads pull them into the 256 MB Infinity
Cache) while the main stream runs cache attention, the o projection and the norm; the streams join before gate|up.  Same kernels
and buffers with and without the side stream, 28 cold weight sets, hipGraph replay.  python tools/micro/mall_prefetch_overlap.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ps_slm_amd.ops import HipOps

ops = HipOps()
M, L, D, I, H, G, ctx = 64, 28, 1536, 8960, 12, 2, 328
HD, LD, W = 128, (H + 2 * G) * 128, G * 128
bf, f32 = torch.bfloat16, torch.float32
rn = lambda *s, k=1.0: (torch.randn(*s, device="cuda") * k).to(bf)
wqkv = [rn(LD, D, k=D ** -0.5) for _ in range(L)]
wo = [rn(D, D, k=D ** -0.5) for _ in range(L)]
wgu = [rn(2 * I, D, k=D ** -0.5) for _ in range(L)]
wd = [rn(D, I, k=I ** -0.5) for _ in range(L)]
bq = rn(LD)
for i in range(L):
    ops.register_decode_weight(wqkv[i], "qkv", LD, H, G)
    ops.register_decode_weight(wo[i], "plain", D)
    ops.register_decode_weight(wgu[i], "swiglu", I)
    ops.register_decode_weight(wd[i], "plain", D, slabs_ok=True)
assert ops.begin_decode(D, D, I)
frag = lambda w: ops._frag[w.data_ptr()][0]
xn, ao, act = (torch.zeros(64, n, dtype=bf, device="cuda") for n in (D, D, I))
x, x2 = torch.randn(M, D, device="cuda"), torch.randn(M, D, device="cuda")
qkv = torch.zeros(M, LD, dtype=bf, device="cuda")
ang = torch.randn(M, 64, device="cuda")
cos, sin = torch.cos(ang), torch.sin(ang)
kc, vc = torch.zeros(L, M * ctx * W, device="cuda", dtype=bf), torch.zeros(L, M * ctx * W, device="cuda", dtype=bf)
pos = torch.full((M,), ctx - 2, device="cuda", dtype=torch.int32)
lens = torch.full((M,), ctx - 1, device="cuda", dtype=torch.int32)
kstart = torch.zeros(M, device="cuda", dtype=torch.int32)
nw = torch.ones(D, device="cuda")
ws = torch.zeros(32 * 64 * 17920, device="cuda")
side = torch.cuda.Stream()
tiny = torch.zeros(64, device="cuda")


def layers(prefetch):
    main = torch.cuda.current_stream()
    for l in range(L):
        ops.gemm_skinny_qkv_rope(xn, wqkv[l], bq, qkv, M, H, G, D, cos, sin, kc[l], vc[l], pos, ctx, ws)
        if prefetch:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                if prefetch == "empty":
                    tiny.zero_()                            # fork / join cost alone
                else:
                    torch.sum(frag(wgu[l]).view(torch.int32), dtype=torch.int32)
                    torch.sum(frag(wd[l]).view(torch.int32), dtype=torch.int32)
        ops.attn_decode(qkv, kc[l], vc[l], None, kstart, lens, ao, M, H, G, ctx, HD ** -0.5)
        ops.gemm_skinny_norm(ao, wo[l], x2, x, M, D, D, nw, xn, 1e-6, ws)
        if prefetch:
            main.wait_stream(side)
        ops.gemm_skinny_swiglu(xn, wgu[l], act, M, I, D, ws)
        ops.gemm_skinny_norm(act, wd[l], x, x2, M, D, I, nw, xn, 1e-6, ws)


for name, pf in (("layers, one stream", False), ("layers + a trivial kernel on a side stream (fork / join)", "empty"),
                 ("layers + Infinity-Cache prefetch on a side stream", True)):
    layers(pf)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        layers(pf)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:52s} {e0.elapsed_time(e1) / 10 / L * 1e3:7.2f} us per layer", flush=True)
