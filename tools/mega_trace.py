"""Where a tasu_decode_layers launch spends its time: per phase, the slowest workgroup's work (barrier exit -> next barrier entry)
and the barrier's own latency (last entry -> median exit).  1.5B geometry, 64 rows.  python tools/mega_trace.py [ctx]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from ps_slm_amd.ops import HipOps
from ps_slm_amd.model import Geometry
from test_gpu_decode_mega import make_case, HD

ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 328
ops = HipOps()
geo = Geometry.from_dict(dict(llm_layers=6))
D, I, H, G, L = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads, geo.llm_layers
M = 64
c = make_case(geo, M, ctx, seed=1)
c["lens"].fill_(ctx - 1); c["slot"].fill_(ctx - 2); c["kstart"].zero_()
for w in c["layers"]:
    ops.register_decode_weight(w["wqkv"], "qkv", w["wqkv"].shape[0], H, G)
    ops.register_decode_weight(w["wo"], "plain", D)
    ops.register_decode_weight(w["wgu"], "swiglu", I)
    ops.register_decode_weight(w["wd"], "plain", D)
assert ops.begin_decode(D, H * HD, I)
same = os.environ.get("MEGA_TRACE_SAME_WEIGHTS") == "1"      # every layer reads layer 0's weights: Infinity-Cache-warm upper bound
table = ops.decode_layer_table([c["layers"][0]] * L if same else c["layers"], c["kc"], c["vc"])
ws = torch.empty(ops.decode_layers_ws_bytes(L, D, H, G, I), dtype=torch.uint8, device="cuda")
xn = torch.zeros(64, D, dtype=torch.bfloat16, device="cuda")
ncu = torch.cuda.get_device_properties(0).multi_processor_count
nbar = 1 + 7 * L
trace = torch.zeros(nbar * ncu * 2, dtype=torch.int64, device="cuda")
run = lambda: ops.decode_layers(table, L, c["x"], c["final_norm"], xn, ws, M, D, H, G, I, c["cos"], c["sin"], c["slot"], c["index"],
                                c["kstart"], c["lens"], ctx, geo.rms_eps, HD ** -0.5)
for _ in range(3):
    run()
torch.cuda.synchronize()
ops.lib.tasu_decode_layers_set_trace(trace.data_ptr(), trace.numel())
run()
torch.cuda.synchronize()
ops.lib.tasu_decode_layers_set_trace(None, 0)
ops.decode_layers_check()
t = trace.cpu().numpy().reshape(nbar, ncu, 2).astype(np.float64) / 100.0       # us
names = ["norm0"] + ["qkv", "attn", "o", "norm", "gate|up", "down", "finish"] * L
print(f"{'phase':10s} {'work max':>9s} {'work med':>9s} {'barrier':>8s}   (us; work = previous barrier exit -> this barrier entry)")
tot_w = tot_b = 0.0
agg = {}
for i in range(1, nbar):
    work = t[i, :, 0] - t[i - 1, :, 1]
    bar = np.median(t[i, :, 1]) - t[i, :, 0].max()
    agg.setdefault(names[i], []).append((work.max(), np.median(work), bar))
for k, v in agg.items():
    v = np.array(v[1:]) if len(v) > 1 else np.array(v)        # skip the first (cold) layer
    print(f"{k:10s} {v[:, 0].mean():9.2f} {v[:, 1].mean():9.2f} {v[:, 2].mean():8.2f}")
    tot_w += v[:, 0].mean(); tot_b += v[:, 2].mean()
print(f"per layer: work {tot_w:.1f} us + barriers {tot_b:.1f} us; whole launch {(t[-1, :, 1].max() - t[0, :, 0].min()):.1f} us for {L} layers")
