"""Times the single-pass attention kernels (csrc/attention_sp.hip) against the tiled ones at the decoder's training shapes:
forward, and the whole backward chain (delta + dQ / dK / dV + rotary backward).  Graph-replayed loops of 28 'layers' over
rotating buffers (cold-ish operands like in the step)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.ops import HipOps

HD = 128
ops = HipOps()
BF = torch.bfloat16


def bench(B, S, H, G, causal=True, layers=28, reps=20):
    M, LD, Spad = B * S, (H + 2 * G) * HD, (S + 63) // 64 * 64
    scale = HD ** -0.5
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = [torch.randn(M, LD, generator=g, device="cuda").to(BF) for _ in range(layers)]
    dout = [torch.randn(M, H * HD, generator=g, device="cuda").to(BF) for _ in range(layers)]
    km = torch.ones(B, Spad, dtype=torch.uint8, device="cuda")
    km[:, S:] = 0
    cos, sin = torch.zeros(M, 64, device="cuda"), torch.zeros(M, 64, device="cuda")
    ops.rope_table(torch.arange(S, dtype=torch.int32, device="cuda").repeat(B), cos, sin, HD, 1e6)
    out = [torch.zeros(M, H * HD, dtype=BF, device="cuda") for _ in range(layers)]
    lse = [torch.zeros(B * H * Spad, device="cuda") for _ in range(layers)]
    delta = torch.zeros(B * H * Spad, device="cuda")
    dqkv = torch.zeros(M, LD, dtype=BF, device="cuda")
    dkp, dvp = torch.zeros(M, H * HD, device="cuda"), torch.zeros(M, H * HD, device="cuda")
    res = {}

    def timed(name, fn):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            gr.replay()
        e1.record(); torch.cuda.synchronize()
        res[name] = round(e0.elapsed_time(e1) / reps / layers * 1e3, 2)      # us per layer

    kernels = ["tiled"] + (["sp"] if ops.lib.tasu_attn_sp_supported(S, H, G) else [])
    for rnd in range(2):                                       # interleaved rounds in one process
        for k in kernels:
            timed(f"fwd_{k}_{rnd}", lambda: [ops.attn_fwd_on(k, qkv[l], km, out[l], lse[l], B, S, H, G, scale, causal) for l in range(layers)])
        for k in kernels + ["gqa"]:
            timed(f"bwd_{k}_{rnd}", lambda: [ops.attn_bwd_fused(qkv[l], km, dout[l], out[l], lse[l], delta, cos, sin, dqkv, dkp, dvp, B, S, H, G,
                                                                scale, causal, k) for l in range(layers)])
    print(json.dumps({"B": B, "S": S, "H": H, "G": G, "causal": causal, "us_per_layer": res}), flush=True)


for shape in ((16, 256, 12, 2), (16, 249, 12, 2), (8, 256, 12, 2), (32, 256, 12, 2), (16, 256, 28, 4)):
    bench(*shape)
