"""Launches the round-4 kernels at their step shapes a few times each (for rocprofv3 --pmc passes: tools/pmc_r04.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps
ops = HipOps()
bf, f32, i32 = torch.bfloat16, torch.float32, torch.int32
M = 4096
rng = torch.tensor([7, 1], dtype=torch.int64, device="cuda")
for K in (1536, 8960):                                            # rank GEMM u = x A^T
    a = torch.randn(M, K, device="cuda").to(bf); b = torch.randn(64, K, device="cuda").to(bf)
    c = torch.empty(M, 64, dtype=bf, device="cuda")
    for _ in range(3): ops.gemm_rank(a, b, c, M, 64, K)
for Mo in (1536, 8960):                                           # weight gradient dB = dy^T u from the row-major dy (K = the step's rows)
    at = torch.randn(M, Mo, device="cuda").to(bf); b = torch.randn(64, M, device="cuda").to(bf)
    c = torch.empty(Mo, 64, dtype=f32, device="cuda")
    for _ in range(3): ops.gemm_rank_tn(at, b, c, Mo, 64, M)
for N in (1536, 8960):                                            # fused accumulate dx += mask . (du A)
    y = torch.randn(M, N, device="cuda").to(bf); u = torch.randn(M, 64, device="cuda").to(bf); w = torch.randn(N, 64, device="cuda").to(bf)
    for _ in range(3): ops.lora_apply(y, u, w, M, N, 64, p=0.05, rng=rng, sid=3)
B, T, V, Kp = 16, 500, 25055, 25088                               # PSD from the logits at SenseVoiceSmall's size
logits = (torch.randn(B * (T + 4), Kp, device="cuda") * 3).to(bf)
lens = torch.full((B,), T, dtype=i32, device="cuda")
fid, fbl, fst = torch.zeros(B * T, dtype=i32, device="cuda"), torch.zeros(B * T, device="cuda"), torch.zeros(B * T, 2, device="cuda")
ss, sl, nl = torch.zeros(B * T, dtype=i32, device="cuda"), torch.zeros(B * T, dtype=i32, device="cuda"), torch.zeros(B, dtype=i32, device="cuda")
body = logits[4:]
for _ in range(3):
    ops.psd_logit_stats(body, lens, fid, fbl, fst, B, T, T + 4, V, 0)
    ops.psd_plan(fid, fbl, lens, ss, sl, nl, B, T, 0, 0.9)
Tout = max(int(nl.max()), 1)
rows = torch.zeros(B * Tout, Kp, device="cuda")
for _ in range(3): ops.psd_gather_softmax(body, fst, ss, sl, nl, rows, B, T, T + 4, Tout, V)
torch.cuda.synchronize()
print("kept rows per utterance", nl.tolist())
