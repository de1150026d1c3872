"""The decode step's layer loop as ONE persistent launch (tasu_decode_layers, csrc/decode_mega.hip) against the same layers
launched one GEMM at a time (ps_slm_amd.decode.layers_per_gemm over tasu_gemm_stream_* / tasu_attn_decode /
tasu_stream_finish_norm): the two paths share their kernel bodies, so the final hidden state and the appended K/V must agree
BIT FOR BIT -- any difference is a synchronisation or visibility bug of the grid barrier, not rounding."""
import numpy as np
import pytest
import torch

from ps_slm_amd.decode import layers_per_gemm
from ps_slm_amd.model import Geometry
from ps_slm_amd.synthetic import MID_GEOMETRY

pytestmark = pytest.mark.gpu
HD = 128
BF, F32, I32 = torch.bfloat16, torch.float32, torch.int32


def from_fragment_order(xf, D):
    """[D/32][4 row tiles][4 lane groups][16 rows][8] -> row-major [64, D]."""
    return xf.view(D // 32, 4, 4, 16, 8).permute(1, 3, 0, 2, 4).reshape(64, D)


def make_case(geo, M, ctx, seed):
    g = torch.Generator().manual_seed(seed)
    D, I, H, G, L = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads, geo.llm_layers
    W = G * HD
    rn = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k)
    layers = []
    for _ in range(L):
        layers.append({
            "wqkv": rn((H + 2 * G) * HD, D, k=D ** -0.5).to(BF).cuda(), "bqkv": rn((H + 2 * G) * HD, k=0.1).to(BF).cuda(),
            "wo": rn(D, H * HD, k=D ** -0.5).to(BF).cuda(), "wgu": rn(2 * I, D, k=D ** -0.5).to(BF).cuda(),
            "wd": rn(D, I, k=I ** -0.5).to(BF).cuda(), "ln1": (1 + rn(D, k=0.1)).cuda(), "ln2": (1 + rn(D, k=0.1)).cuda()})
    final_norm = (1 + rn(D, k=0.1)).cuda()
    rs = np.random.RandomState(seed)
    lens = rs.randint(ctx // 2, ctx + 1, size=M).astype(np.int32)
    kstart = rs.randint(0, 12, size=M).astype(np.int32)
    slot = lens - 1
    index = rs.randint(0, M, size=(M, ctx)).astype(np.int32)
    index[np.arange(M), slot] = np.arange(M)                     # a row appends into itself
    ang = rn(M, HD // 2, k=3.0)
    return dict(layers=layers, final_norm=final_norm, x=rn(M, D).cuda(), cos=torch.cos(ang).cuda(), sin=torch.sin(ang).cuda(),
                kc=rn(L, M * ctx * W, k=0.5).to(BF).cuda(), vc=rn(L, M * ctx * W, k=0.5).to(BF).cuda(),
                index=torch.from_numpy(index).cuda(), kstart=torch.from_numpy(kstart).cuda(), slot=torch.from_numpy(slot).cuda(),
                lens=torch.from_numpy(lens).cuda())


@pytest.mark.parametrize("size,M,ctx", [("mid", 64, 96), ("mid", 37, 160), ("mid", 5, 64), ("1.5b", 64, 200), ("1.5b", 24, 330)])
def test_one_launch_equals_per_gemm_launches_bit_for_bit(size, M, ctx, monkeypatch):
    from ps_slm_amd.ops import HipOps
    monkeypatch.setenv("TASU_DECODE_DOWN_SLABS", "1")            # the per-GEMM reference sums the same K-range slabs
    ops = HipOps()
    ops.use_mega = True                                          # (experimental path: off by default)
    geo = Geometry.from_dict(dict(MID_GEOMETRY, llm_layers=3)) if size == "mid" else Geometry.from_dict(dict(llm_layers=3))
    D, I, H, G, L = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads, geo.llm_layers
    c = make_case(geo, M, ctx, seed=M + ctx)
    for w in c["layers"]:
        ops.register_decode_weight(w["wqkv"], "qkv", w["wqkv"].shape[0], H, G)
        ops.register_decode_weight(w["wo"], "plain", D)
        ops.register_decode_weight(w["wgu"], "swiglu", I)
        ops.register_decode_weight(w["wd"], "plain", D)
    assert ops.begin_decode(D, H * HD, I) and ops.dec_frag_act
    assert ops.decode_layers_supported(M, D, H, G, I, ctx)
    z = lambda *s, dt=BF: torch.zeros(*s, dtype=dt, device="cuda")
    # ---- reference: one launch per GEMM
    kc_a, vc_a = c["kc"].clone(), c["vc"].clone()
    xn_a, x_a = z(64, D), c["x"].clone()
    layers_per_gemm(ops, geo, c["layers"], c["final_norm"], x_a, z(M, D, dt=F32), xn_a, z(M, (H + 2 * G) * HD), z(64, H * HD),
                    z(64, I), c["cos"], c["sin"], kc_a, vc_a, c["index"], c["kstart"], c["slot"], c["lens"], M, ctx,
                    z(8 * 64 * max(D, 2 * I), dt=F32))
    # ---- one persistent launch (twice: the barrier state carries over from launch to launch)
    table_kc, table_vc = c["kc"].clone(), c["vc"].clone()
    table = ops.decode_layer_table(c["layers"], table_kc, table_vc)
    ws = torch.empty(ops.decode_layers_ws_bytes(L, D, H, G, I), dtype=torch.uint8, device="cuda")
    assert ws.data_ptr() % 256 == 0
    for rep in range(2):
        table_kc.copy_(c["kc"]), table_vc.copy_(c["vc"])
        ws.fill_(0xFF)                                           # NaN patterns: a read of a not-yet-written intermediate shows
        xn_b = z(64, D)
        ops.decode_layers(table, L, c["x"], c["final_norm"], xn_b, ws, M, D, H, G, I, c["cos"], c["sin"], c["slot"], c["index"],
                          c["kstart"], c["lens"], ctx, geo.rms_eps, HD ** -0.5)
        torch.cuda.synchronize()
        ops.decode_layers_check()
        a = from_fragment_order(xn_a, D)[:M].view(torch.int16)
        b = from_fragment_order(xn_b, D)[:M].view(torch.int16)
        assert torch.isfinite(from_fragment_order(xn_b, D)[:M].float()).all()
        assert torch.equal(a, b), (rep, int((a != b).sum()), a.numel())
        assert torch.equal(kc_a.view(torch.int16), table_kc.view(torch.int16))
        assert torch.equal(vc_a.view(torch.int16), table_vc.view(torch.int16))
    ops.end_decode()
