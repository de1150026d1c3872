"""Tile timeline of workgroup 0 of the streaming decode GEMMs from the instrumented build (make -C ps_slm_amd/csrc trace;
TASU_LIB_PATH=ps_slm_amd/libtasu_hip_trace.so).  1.5B shapes, 64 rows, 28 rotating (cold) weight sets."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ps_slm_amd.ops import HipOps

ops = HipOps()
M, L, D, I, H, G = 64, 28, 1536, 8960, 12, 2
bf = torch.bfloat16
rn = lambda *s, k=1.0: (torch.randn(*s, device="cuda") * k).to(bf)
wgu = [rn(2 * I, D, k=D ** -0.5) for _ in range(L)]
wd = [rn(D, I, k=I ** -0.5) for _ in range(L)]
wo = [rn(D, D, k=D ** -0.5) for _ in range(L)]
for i in range(L):
    ops.register_decode_weight(wgu[i], "swiglu", I)
    ops.register_decode_weight(wd[i], "plain", D, slabs_ok=True)
    ops.register_decode_weight(wo[i], "plain", D)
assert ops.begin_decode(D, D, I)
xn, act, ao = rn(M, D), rn(M, I), rn(M, D)
ws = torch.zeros(32 * 64 * 17920, device="cuda")
x, x2 = torch.randn(M, D, device="cuda"), torch.randn(M, D, device="cuda")
nw = torch.ones(D, device="cuda")
y = torch.zeros(M, D, device="cuda", dtype=bf)


def trace(name, fn, nstamp):
    acc, n = np.zeros(nstamp), 0
    for it in range(3 * L):
        fn(it % L)
        torch.cuda.synchronize()
        buf = (ctypes.c_uint64 * 32)()
        assert ops.lib.tasu_stream_trace_read(buf) == 0
        t = np.array(buf[:nstamp], dtype=np.float64) / 100.0
        if it >= L:
            acc += t - t[0]
            n += 1
    acc /= n
    print(name, " ".join(f"{v:6.2f}" for v in acc[1:]), "(us after the start: loads issued, then after every tile)")


trace("gate|up (5 tiles, ring)", lambda i: ops.gemm_skinny_swiglu(xn, wgu[i], act, M, I, D, ws), 8)
trace("down slabs (4 tiles)   ", lambda i: ops.gemm_skinny_norm(act, wd[i], x, x2, M, D, I, nw, y, 1e-6, ws), 8)
trace("o (1 tile)             ", lambda i: ops.gemm_skinny_norm(ao, wo[i], x2, x, M, D, D, nw, y, 1e-6, ws), 4)
