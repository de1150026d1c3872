"""Micro-benchmark of the attention kernels at the training shape (B=16, S=256, H=12, G=2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps
ops = HipOps()
B, S, H, G, HD = 16, 256, 12, 2, 128
M, LD, Spad = B * S, (H + 2 * G) * HD, 256
bf, f32 = torch.bfloat16, torch.float32
qkv = torch.randn(M, LD, device="cuda").to(bf)
km = torch.ones(B, Spad, dtype=torch.uint8, device="cuda")
cos, sin = torch.ones(M, 64, device="cuda"), torch.zeros(M, 64, device="cuda")
qt, kt, vt = (torch.zeros(B * n * HD * Spad, dtype=bf, device="cuda") for n in (H, G, G))
ops.rope_fwd(qkv, cos, sin, qt, kt, vt, B, S, H, G)
out = torch.zeros(M, H * HD, dtype=bf, device="cuda"); lse = torch.zeros(B * H * Spad, device="cuda")
dout = torch.randn(M, H * HD, device="cuda").to(bf); delta = torch.zeros(B * H * Spad, device="cuda")
dout_t = torch.zeros(B * H * HD * Spad, dtype=bf, device="cuda"); dqkv = torch.zeros(M, LD, dtype=bf, device="cuda")
dkp, dvp = torch.zeros(M, H * HD, device="cuda"), torch.zeros(M, H * HD, device="cuda")
sc = HD ** -0.5
calls = {"fwd": lambda: ops.attn_fwd(qkv, vt, km, out, lse, B, S, H, G, sc, True),
         "prep": lambda: ops.attn_bwd_prep(dout, out, delta, dout_t, B, S, H),
         "dq": lambda: ops.attn_bwd_dq(qkv, kt, km, dout, lse, delta, dqkv, B, S, H, G, sc, True),
         "dkv": lambda: ops.attn_bwd_dkv(qkv, qt, km, dout, dout_t, lse, delta, dkp, dvp, B, S, H, G, sc, True),
         "bwd (dq+dkv, one launch)": lambda: ops.attn_bwd(qkv, qt, kt, km, dout, dout_t, lse, delta, dqkv, dkp, dvp, B, S, H, G, sc, True),
         "rope_bwd": lambda: ops.rope_bwd(dqkv, dkp, dvp, cos, sin, B, S, H, G)}
for name, fn in calls.items():
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:26s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
