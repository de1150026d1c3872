"""Does a weight matrix that was just READ by another kernel stream faster into the decode GEMM that follows (Infinity Cache hit)
than one coming cold from HBM?  28 rotating weight sets (cold), hipGraph replay; the 'read' is a plain torch reduction over
the same bytes.  python tools/micro/mall_prefetch.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ps_slm_amd.ops import HipOps

ops = HipOps()
M, L, D, I = 64, 28, 1536, 8960
bf = torch.bfloat16
rn = lambda *s, k=1.0: (torch.randn(*s, device="cuda") * k).to(bf)
wgu = [rn(2 * I, D, k=D ** -0.5) for _ in range(L)]
wd = [rn(D, I, k=I ** -0.5) for _ in range(L)]
for i in range(L):
    ops.register_decode_weight(wgu[i], "swiglu", I)
    ops.register_decode_weight(wd[i], "plain", D, slabs_ok=True)
assert ops.begin_decode(D, D, I)
xn, act = rn(M, D), rn(M, I)
ws = torch.zeros(32 * 64 * 17920, device="cuda")
x, x2 = torch.randn(M, D, device="cuda"), torch.randn(M, D, device="cuda")
nw = torch.ones(D, device="cuda")
y = torch.zeros(M, D, device="cuda", dtype=bf)
sink = torch.zeros(L, device="cuda")
frag = lambda w: ops._frag[w.data_ptr()][0]


def timed(name, fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 / L * 1e3
    print(f"{name:48s} {us:7.2f} us per layer", flush=True)
    return us


read_gu = lambda i: torch.sum(frag(wgu[i]).view(torch.int32), dtype=torch.int32)
a = timed("gate|up alone (cold)", lambda: [ops.gemm_skinny_swiglu(xn, wgu[i], act, M, I, D, ws) for i in range(L)])
b = timed("read of the gate|up weights alone", lambda: [read_gu(i) for i in range(L)])
c = timed("read, then gate|up", lambda: [(read_gu(i), ops.gemm_skinny_swiglu(xn, wgu[i], act, M, I, D, ws)) for i in range(L)])
print(f"  -> gate|up after its weights were just read: {c - b:.2f} us (cold {a:.2f})")
read_d = lambda i: torch.sum(wd[i].view(torch.int32), dtype=torch.int32)
a = timed("down alone (cold)", lambda: [ops.gemm_skinny_norm(act, wd[i], x, x2, M, D, I, nw, y, 1e-6, ws) for i in range(L)])
b = timed("read of the down weights alone", lambda: [read_d(i) for i in range(L)])
c = timed("read, then down", lambda: [(read_d(i), ops.gemm_skinny_norm(act, wd[i], x, x2, M, D, I, nw, y, 1e-6, ws)) for i in range(L)])
print(f"  -> down after its weights were just read: {c - b:.2f} us (cold {a:.2f})")
