"""fp32 decode-step GEMMs alone, on rotating (cold) weight copies: microseconds and weight TB/s per shape.

    python tools/bench_f32_stream.py [rows]          # rows: beam rows (default 64)
The switches below exist in the LAB build (make -C ps_slm_amd/csrc lab; TASU_LIB_PATH=ps_slm_amd/libtasu_hip_lab.so):
    TASU_F32_STREAM=0 python tools/bench_f32_stream.py     # the tile kernel (csrc/fp32.hip f32_gemm_kernel) on the same shapes
    TASU_F32_STREAM_KS=3 python tools/bench_f32_stream.py  # force the streaming kernel's K slice (ksplit = K / (128 KS))

Shapes: the four projections of a Qwen2.5-1.5B layer and the lm_head, through the entry points the decode loop calls
(tasu_f32_gemm_qkv_rope / _resid_rmsnorm / _swiglu / _nt), finishers included."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ps_slm_amd.ops import HipOps  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
D, I, H, G, HD, V = 1536, 8960, 12, 2, 128, 151936
dev = "cuda"
ops = HipOps()
f32 = torch.float32
g = torch.Generator(device=dev).manual_seed(1)


def rnd(*shape, scale=0.02):
    return torch.randn(*shape, device=dev, dtype=f32, generator=g) * scale


ws = torch.empty(max(16 * 128 * 4096, 2 * 64 * V), device=dev, dtype=f32)
x = rnd(M, D, scale=1.0)
xn = rnd(M, D, scale=1.0)
ao = rnd(M, H * HD, scale=1.0)
act = rnd(M, I, scale=1.0)
qkv = torch.empty(M, (H + 2 * G) * HD, device=dev, dtype=f32)
gu = torch.empty(M, 2 * I, device=dev, dtype=f32)
logits = torch.empty(M, V, device=dev, dtype=f32)
cos = torch.ones(M, HD, device=dev, dtype=f32)
sin = torch.zeros(M, HD, device=dev, dtype=f32)
ln = torch.ones(D, device=dev, dtype=f32)
bq = rnd((H + 2 * G) * HD)


def copies(n, k, total_mb=600):
    c = max(2, int(total_mb * 1e6 / (n * k * 4)) + 1)
    return [rnd(n, k) for _ in range(c)]


def timed(name, ws_list, fn, nbytes, reps=40):
    for w in ws_list[:2]:
        fn(w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        fn(ws_list[r % len(ws_list)])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return {"shape": name, "us": round(us, 2), "weight_TBps": round(nbytes / us / 1e6, 3)}


out = []
w = copies((H + 2 * G) * HD, D)
out.append(timed("qkv+rope", w, lambda t: ops.f32_gemm_qkv_rope(xn, t, bq, qkv, cos, sin, M, H, G, D, ws), w[0].numel() * 4))
w = copies(D, H * HD)
out.append(timed("o+resid+norm", w, lambda t: ops.f32_gemm_resid_rmsnorm(ao, t, x, ln, xn, M, D, H * HD, 1e-6, ws, resid=x), w[0].numel() * 4))
w = copies(2 * I, D)
out.append(timed("gate|up+swiglu", w, lambda t: ops.f32_gemm_swiglu(xn, t, gu, act, M, I, D, ws), w[0].numel() * 4))
w = copies(D, I)
out.append(timed("down+resid+norm", w, lambda t: ops.f32_gemm_resid_rmsnorm(act, t, x, ln, xn, M, D, I, 1e-6, ws, resid=x), w[0].numel() * 4))
del w
torch.cuda.empty_cache()
w = copies(V, D, total_mb=1000)
out.append(timed("lm_head", w, lambda t: ops.f32_gemm(xn, t, logits, M, V, D, ws=ws), w[0].numel() * 4, reps=12))
layer = sum(r["us"] for r in out[:4])
print(json.dumps({"rows": M, "env": {k: v for k, v in os.environ.items() if k.startswith("TASU_F32")}, "shapes": out,
                  "layer_us": round(layer, 1), "position_ms_gemms": round((28 * layer + out[4]["us"]) / 1e3, 3)}))
