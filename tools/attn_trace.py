"""Phase timeline of tasu_attn_decode's workgroup (0, 0) from the instrumented build (-DTASU_ATTN_TRACE, loaded through
TASU_LIB_PATH): see the build recipe in DESIGN.md 4c.  1.5B geometry, 64 rows, context 328."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ps_slm_amd.ops import HipOps

ops = HipOps()
M, H, G, ctx, HD = 64, 12, 2, 328, 128
W, LD = G * HD, (H + 2 * G) * HD
bf = torch.bfloat16
qkv = torch.randn(M, LD, device="cuda").to(bf)
kc = torch.randn(28, M * ctx * W, device="cuda").to(bf)
vc = torch.randn(28, M * ctx * W, device="cuda").to(bf)
index = torch.arange(M, dtype=torch.int32, device="cuda").repeat_interleave(ctx).view(M, ctx).contiguous()
kstart = torch.zeros(M, dtype=torch.int32, device="cuda")
lens = torch.full((M,), 228, dtype=torch.int32, device="cuda")
out = torch.zeros(M, H * HD, device="cuda", dtype=bf)
names = ["start", "index in LDS", "K/V/q loads issued", "scores done", "sync", "softmax done", "P.V done", "partials in LDS", "stored"]
acc = np.zeros(9)
n = 0
for it in range(40):
    l = it % 28
    ops.attn_decode(qkv, kc[l], vc[l], index, kstart, lens, out, M, H, G, ctx, HD ** -0.5)
    torch.cuda.synchronize()
    buf = (ctypes.c_uint64 * 16)()
    assert ops.lib.tasu_attn_trace_read(buf) == 0
    t = np.array(buf[:9], dtype=np.float64) / 100.0
    if it >= 12:
        acc += t - t[0]
        n += 1
acc /= n
for i in range(1, 9):
    print(f"{names[i]:22s} +{acc[i] - acc[i - 1]:6.2f} us   (at {acc[i]:6.2f})")
