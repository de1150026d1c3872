"""LoRA recipe on the GPU: the elementwise kernels of csrc/lora.hip bit-exact against the CPU double, the adapted training step
(HipOps through the C-ABI) against the double and against the reference goldens of oracle/make_golden_lora.py."""
import numpy as np
import pytest
import torch

from fake_ops import FakeOps
from test_lora_cpu import build, check_against_golden, cosine, golden_case, run_text

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from ps_slm_amd.ops import HipOps
    return HipOps()


def test_lora_elementwise_kernels_bit_exact(ops):
    f = FakeOps()
    g = torch.Generator().manual_seed(1)
    M, D = 37, 256
    n = M * D
    y = torch.randn(n, generator=g).bfloat16()
    for s in (0.25, 2.0, 0.3):
        dc, dg = torch.empty_like(y), torch.empty(n, dtype=torch.bfloat16, device="cuda")
        f.scale_bf16(y, dc, s)
        ops.scale_bf16(y.cuda(), dg, s)
        assert torch.equal(dg.cpu(), dc)
    rng = torch.tensor([123456789, 4], dtype=torch.int64)
    rng_g = rng.cuda()
    for p, sid in ((0.05, 0), (0.25, 13), (0.5, 221)):
        dc, dg = torch.empty_like(y), torch.empty(n, dtype=torch.bfloat16, device="cuda")
        f.lora_dropout(y, dc, p, rng, sid)
        ops.lora_dropout(y.cuda(), dg, p, rng_g, sid)
        assert torch.equal(dg.cpu(), dc)
        assert abs(float((dc == 0).float().mean()) - p) < 0.02
        x = torch.randn(M, D, generator=g)
        w = 1 + 0.1 * torch.randn(D, generator=g)
        rstd = torch.rsqrt(x.pow(2).mean(-1) + 1e-6)
        dc2, dg2 = torch.empty(M, D, dtype=torch.bfloat16), torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
        f.lora_dropout_norm(x, w, rstd, dc2, M, D, p, rng, sid)
        ops.lora_dropout_norm(x.cuda(), w.cuda(), rstd.cuda(), dg2, M, D, p, rng_g, sid)
        assert torch.equal(dg2.cpu(), dc2)
    # a column slice of a wider buffer as the source (the SwiGLU output inside the down projection's K-extended operand)
    wide = torch.randn(M, D + 64, generator=g).bfloat16()
    dc3, dg3 = torch.empty(M, D, dtype=torch.bfloat16), torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    f.lora_dropout(wide[:, :D], dc3, 0.25, rng, 9)
    ops.lora_dropout(wide.cuda()[:, :D], dg3, 0.25, rng_g, 9)
    assert torch.equal(dg3.cpu(), dc3)
    ops.rng_advance(rng_g)
    assert rng_g.cpu().tolist() == [123456789, 5]
    # a different step or stream draws a different mask; the same (seed, step, stream) the same one
    a, b, c = (torch.empty(n, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    ones = torch.ones(n, dtype=torch.bfloat16, device="cuda")
    ops.lora_dropout(ones, a, 0.5, rng_g, 3)
    ops.lora_dropout(ones, b, 0.5, rng_g, 3)
    ops.lora_dropout(ones, c, 0.5, rng_g, 4)
    assert torch.equal(a, b) and not torch.equal(a, c)


@pytest.mark.parametrize("M,n_members,drop", [(4096, 3, True), (100, 2, False), (64, 1, True), (333, 4, True)])
def test_lora_group_launches_have_the_bits_of_the_member_launches(ops, M, n_members, drop):
    """One launch per adapted GROUP (round 6) against one per member: the members' rank GEMMs with their own operands and K
    (tasu_gemm_nt_rank_group), the fused accumulate applied member after member in one pass over y (tasu_lora_apply_group) and
    the dropped norm copies from one evaluation of the norm (tasu_lora_dropout_norm_group) -- bit-identical to the loops."""
    g = torch.Generator().manual_seed(7)
    dev = "cuda"
    rnd = lambda *sh, scale=1.0: (torch.randn(*sh, generator=g) * scale).to(dev)
    rng = torch.tensor([987654321, 3], dtype=torch.int64, device=dev)
    p = 0.1 if drop else 0.0
    sids = [5 + 8 * t for t in range(n_members)]
    # rank GEMMs: members read column slices of one wide matrix (the q|k|v / gate|up gradient), each with its own K
    Ks = [256, 64, 512, 128][:n_members]
    wide = rnd(M, sum(Ks) + 16).bfloat16()
    offs = np.cumsum([0] + Ks)
    As = [wide[:, offs[t]:offs[t] + Ks[t]] for t in range(n_members)]
    Bs = [rnd(64, Ks[t], scale=Ks[t] ** -0.5).bfloat16() for t in range(n_members)]
    du_g, du_l = torch.zeros(M, n_members * 64, dtype=torch.bfloat16, device=dev), torch.zeros(M, n_members * 64, dtype=torch.bfloat16, device=dev)
    ops.gemm_rank_group(As, Bs, [du_g[:, 64 * t:64 * t + 64] for t in range(n_members)], M, 64, Ks)
    for t in range(n_members):
        ops.gemm_rank(As[t], Bs[t], du_l[:, 64 * t:64 * t + 64], M, 64, Ks[t])
    torch.cuda.synchronize()
    assert torch.equal(du_g, du_l) and float(du_g.float().abs().max()) > 0
    # accumulate: y += mask_t . bf16(s . bf16(du_t W_t^T)), member after member
    N = 520
    Ws = [rnd(N, 64, scale=0.2).bfloat16() for _ in range(n_members)]
    y0 = rnd(M, N + 8).bfloat16()
    yg, yl = y0.clone(), y0.clone()
    us = [du_g[:, 64 * t:64 * t + 64] for t in range(n_members)]
    ops.lora_apply_group(yg[:, :N], us, Ws, M, N, 64, sids, s=0.25, p=p, rng=rng)
    for t in range(n_members):
        ops.lora_apply(yl[:, :N], us[t], Ws[t], M, N, 64, s=0.25, p=p, rng=rng, sid=sids[t])
    torch.cuda.synchronize()
    assert torch.equal(yg, yl) and not torch.equal(yg, y0) and torch.equal(yg[:, N:], y0[:, N:])
    # dropped norm copies
    if drop:
        D = 512
        x, w = rnd(M, D), 1 + 0.1 * rnd(D)
        rstd = torch.rsqrt(x.pow(2).mean(-1) + 1e-6)
        dg = [torch.empty(M, D, dtype=torch.bfloat16, device=dev) for _ in range(n_members)]
        dl = [torch.empty(M, D, dtype=torch.bfloat16, device=dev) for _ in range(n_members)]
        ops.lora_dropout_norm_group(x, w, rstd, dg, M, D, p, rng, sids)
        for t in range(n_members):
            ops.lora_dropout_norm(x, w, rstd, dl[t], M, D, p, rng, sids[t])
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(dg, dl))
        assert n_members == 1 or not torch.equal(dg[0], dg[1])      # every member its own mask


@pytest.mark.parametrize("name", ["mid_text_lora", "mid_text_lora_qv", "mid_text_lora_drop"])
def test_lora_step_hip_vs_double_and_reference_golden(ops, name):
    z, geo, cfg, sd, lsd, batch = golden_case(name)
    rng = z["rng"] if cfg.lora_dropout > 0 else None
    gm = build(geo, cfg, sd, lsd, ops, "cuda", rng)
    cm = build(geo, cfg, sd, lsd, FakeOps(), "cpu", rng)
    sg, sc = run_text(gm, batch), run_text(cm, batch)
    torch.cuda.synchronize()
    lg, lc = sg.dev["loss_out"].cpu(), sc.dev["loss_out"]
    assert abs(float(lg[0]) - float(lc[0])) < 2e-3
    valid = torch.from_numpy(sc.plan.key_mask[:, : sc.S].astype(bool))
    a, b = gm.logits_view(sg).float().cpu()[valid], cm.logits_view(sc).float()[valid]
    assert float((a - b).abs().max() / b.abs().max()) < 2e-2
    for (k, g1), (_, g2) in zip(sorted(gm.lora_grads().items()), sorted(cm.lora_grads().items())):
        assert cosine(g1, g2) > 0.9995, k
        assert float((g1.cpu() - g2).norm() / g2.norm()) < 3e-2, k
    for k, g2 in cm.projector_grads().items():
        assert cosine(gm.projector_grads()[k], g2) > 0.9995, k
    check_against_golden(gm, sg, z)


def test_generate_with_adapters_on_gpu(ops):
    """Beam-4 decode of the adapted model on the HIP kernels (merged weights through the streaming decode path) against the
    hand-LoRA reference's tokens (common prefix) and the CPU double's."""
    from test_lora_cpu import gen_inputs, generate_text
    geo, cfg, sd, lsd, ids, am, word_ids, ref, ref_base = gen_inputs()
    gm = build(geo, cfg, sd, lsd, ops, "cuda")
    cm = build(geo, cfg, sd, lsd, FakeOps(), "cpu")
    tg, tc = generate_text(gm, ids, am, word_ids), generate_text(cm, ids, am, word_ids)
    assert ((tg == ref).cumprod(1).sum(1) >= 8).all(), (tg, ref)
    assert ((tg == tc).cumprod(1).sum(1) >= 8).all(), (tg, tc)
    # the adapted training step still runs on the un-merged weights afterwards
    assert gm._lora_run is not None and gm.llm is not gm.lora._merged


def test_generate_with_adapters_margin_cases_exact_on_gpu(ops):
    """token ids are index work.  On the 7 rounding-stable decode cases of the ADAPTED model
    (tests/golden/mid_generate_lora_margin.npz: reference fp32 with hand-applied LoRA == bf16 oracle on the merged weights == 8
    jittered runs == the CPU double) the HIP decode path on the merged weights (hi/lo-split MFMA merge, streaming decode kernels,
    beam kernels) must EQUAL the reference's tokens."""
    from conftest import decode_lora_margin_cases
    from test_lora_cpu import decode_lora_margin
    geo, cfg, sd, lsd, cases = decode_lora_margin_cases()
    gm = build(geo, cfg, sd, lsd, ops, "cuda")
    assert not decode_lora_margin(gm, geo, cases)


def test_lora_step_at_benchmark_shape(ops):
    """The use_peft=true step bench.py --lora times: Qwen2.5-1.5B geometry, 16 utterances x S = 256, r = 64 on all seven Linears,
    dropout 0.05.  Properties: peft's zero-B start (dA == 0 exactly, dB != 0, loss = the frozen model's); with non-zero adapters
    bitwise determinism for a fixed mask step, hipGraph replay == eager launches, new masks on every replay of the step's graph."""
    from ps_slm_amd.lora import LoraConfig
    from ps_slm_amd.model import Geometry, TasuModel
    from ps_slm_amd.synthetic import random_lora_state_dict, synthetic_text_batch
    geo = Geometry.qwen25_1p5b()
    m = TasuModel(geo, ops, "cuda", keep_logits=False)
    m.init_random(seed=1234)
    cfg = LoraConfig(r=64, lora_alpha=16, lora_dropout=0.05)
    m.enable_lora(cfg, seed=11)
    lp = m.lora
    batch = synthetic_text_batch(geo, 16, seed=1234, noise=False)

    def train_step(graphs=False, step=None):
        m.use_graphs = graphs
        if step is not None:
            lp.seed_dropout(11, step)
        st = m.prepare_text(batch["input_ids"], batch["attention_mask"], batch["labels"], batch["post_ids"])
        m.run_forward_text(st)
        m.run_backward(st)
        torch.cuda.synchronize()
        m.use_graphs = False
        return st, st.dev["loss_out"].clone(), m.proj.g.clone()

    st, loss0, g0 = train_step(step=0)
    assert st.M == 4096 and st.S == 256 and torch.isfinite(loss0).all() and torch.isfinite(g0).all()
    for key, k in lp.names():
        gv = lp.view(g0, *k)
        assert (float(gv.abs().max()) == 0.0) == ("lora_A" in key), key
    m.lora, m._lora_run, keep = None, None, (m.lora, m._lora_run)
    try:                                                       # the frozen recipe on the same weights: same loss (B = 0 adds nothing)
        stf = m.prepare_text(batch["input_ids"], batch["attention_mask"], batch["labels"], batch["post_ids"])
        m.tail_rows = False
        m.run_forward_text(stf)
        torch.cuda.synchronize()
        assert abs(float(stf.dev["loss_out"][0]) - float(loss0[0])) < 2e-5
    finally:
        m.lora, m._lora_run = keep
        m.tail_rows = True
    lp.load_state_dict(random_lora_state_dict(geo, cfg, 5, b_scale=0.02))
    m.sync_projector_copies()
    _, l1, g1 = train_step(step=3)
    _, l2, g2 = train_step(step=3)
    assert torch.equal(l1, l2) and torch.equal(g1, g2)                                   # same mask step: bitwise repeatable
    _, l3, g3 = train_step(step=4)
    assert not torch.equal(l1, l3)                                                       # another step: other masks
    outs = []
    lp.seed_dropout(11, 3)
    for i in range(4):                                                                   # eager warm-up, capture, replays
        outs.append(train_step(graphs=True)[1:])
    assert int(lp.rng[1]) == 7                                                           # every launch of the step advanced the mask step
    assert torch.equal(outs[0][0], l1) and torch.equal(outs[0][1], g1)                   # eager, masks of step 4
    assert torch.equal(outs[1][0], l3) and torch.equal(outs[1][1], g3)                   # captured + replayed, masks of step 5
    _, l6, g6 = train_step(step=6)
    assert torch.equal(outs[3][0], l6) and torch.equal(outs[3][1], g6)                   # a later replay == eager at the same mask step
    # round 6: the adapted BACKWARD is launched eagerly (its side-stream forks, captured, ran at 82 ms as a process's first workload);
    # the forward is the graph that was captured above
    tags = [k[0] if not isinstance(k[0], tuple) else k[0][0] for k in m._graphs]
    assert any(str(t).startswith("fwd") for t in tags) and not any(str(t).startswith("bwd") for t in tags), tags
    assert m._lora_run.side is not None and not m.lora_bwd_graphs


@pytest.mark.parametrize("M,N,K", [(4096, 64, 1536), (4096, 64, 8960), (1536, 64, 4096), (8960, 64, 4096), (141, 16, 256), (333, 24, 192),
                                   (64, 8, 64), (2048, 48, 512), (17, 64, 1024)])
def test_gemm_rank_vs_fp32(ops, M, N, K):
    """csrc/gemm_rank.hip (the adapters' rank-sized GEMMs) against an fp32 product of the same bf16 operands: bf16 / fp32 output,
    plain / transposed store, strided operands and outputs (column slices of wider buffers), nothing written outside C."""
    g = torch.Generator().manual_seed(M + N + K)
    a_full = (torch.randn(M, K + 64, generator=g) * 0.5).bfloat16().cuda()
    b_full = (torch.randn(N, K + 8, generator=g) * 0.5).bfloat16().cuda()
    a, b = a_full[:, 64:], b_full[:, :K]
    ref = a.float() @ b.float().t()
    tol = 2e-3 * float(ref.abs().max()) + 1e-6
    c = torch.full((M, N + 16), 7.0, dtype=torch.float32, device="cuda")
    ops.gemm_rank(a, b, c[:, 8:8 + N], M, N, K, f32=True)
    assert float((c[:, 8:8 + N] - ref).abs().max()) < tol
    assert float((c[:, :8] - 7).abs().max()) == 0 and float((c[:, 8 + N:] - 7).abs().max()) == 0      # nothing outside the N columns
    cb = torch.zeros(M, N + 8, dtype=torch.bfloat16, device="cuda")
    ops.gemm_rank(a, b, cb, M, N, K)
    assert torch.equal(cb[:, :N], c[:, 8:8 + N].bfloat16()) and float(cb[:, N:].abs().max()) == 0
    ct = torch.full((N, M + 3), 7.0, dtype=torch.float32, device="cuda")
    ops.gemm_rank(a, b, ct, M, N, K, f32=True, transposed=True)
    assert torch.equal(ct[:, :M], c[:, 8:8 + N].t()) and float((ct[:, M:] - 7).abs().max()) == 0
    c2 = torch.empty(M, N, dtype=torch.float32, device="cuda")
    ops.gemm_rank(a, b, c2, M, N, K, f32=True)
    assert torch.equal(c2, c[:, 8:8 + N])                                                              # deterministic


@pytest.mark.parametrize("M,N,K", [(1536, 64, 4096), (8960, 64, 4096), (256, 64, 4096), (64, 16, 64), (1536, 64, 1984), (512, 40, 640), (128, 64, 512)])
def test_gemm_rank_tn_vs_fp32_and_the_transposed_path(ops, M, N, K):
    """tasu_gemm_tn_rank (the adapters' weight gradients with the big operand K-major as the step leaves it: hardware transpose
    reads) against an fp32 product and against tasu_gemm_nt_rank on a transposed copy (same chunks, same wave order: equal up to
    the order of the 32 products inside one MFMA); strided operands (column slices), plain / transposed store, nothing outside C."""
    g = torch.Generator().manual_seed(M + N + K)
    at_full = (torch.randn(K, M + 128, generator=g) * 0.5).bfloat16().cuda()
    b_full = (torch.randn(N, K + 8, generator=g) * 0.5).bfloat16().cuda()
    at, b = at_full[:, 64:64 + M], b_full[:, :K]
    ref = at.float().t() @ b.float().t()
    tol = 2e-3 * float(ref.abs().max()) + 1e-6
    c = torch.full((M, N + 16), 7.0, dtype=torch.float32, device="cuda")
    ops.gemm_rank_tn(at, b, c[:, 8:8 + N], M, N, K)
    assert float((c[:, 8:8 + N] - ref).abs().max()) < tol
    assert float((c[:, :8] - 7).abs().max()) == 0 and float((c[:, 8 + N:] - 7).abs().max()) == 0
    nt = torch.empty(M, N, dtype=torch.float32, device="cuda")
    ops.gemm_rank(at.t().contiguous(), b, nt, M, N, K, f32=True)
    assert float((c[:, 8:8 + N] - nt).abs().max()) <= 1e-5 * float(ref.abs().max())
    ct = torch.full((N, M + 3), 7.0, dtype=torch.float32, device="cuda")
    ops.gemm_rank_tn(at, b, ct, M, N, K, transposed=True)
    assert torch.equal(ct[:, :M], c[:, 8:8 + N].t()) and float((ct[:, M:] - 7).abs().max()) == 0
    c2 = torch.empty(M, N, dtype=torch.float32, device="cuda")
    ops.gemm_rank_tn(at, b, c2, M, N, K)
    assert torch.equal(c2, c[:, 8:8 + N])                                                              # deterministic


@pytest.mark.parametrize("M,N,R", [(4096, 1536, 64), (333, 512, 64), (64, 128, 128), (141, 256, 64), (200, 8960, 64), (70, 264, 64)])
def test_lora_apply_bit_exact_vs_double(ops, M, N, R):
    """tasu_lora_apply (rank-R product + scale + dropout mask + accumulate + residual add in one pass over y) against the CPU
    double: the v image is bf16 of an fp32 MFMA sum, so the comparison allows the last-place flips of that sum's order and is
    bit-exact everywhere else; strided y / u / W (column slices), nothing written outside [M, N]."""
    f = FakeOps()
    g = torch.Generator().manual_seed(M + N)
    y_full = torch.randn(M, N + 16, generator=g).bfloat16()
    u_full = (torch.randn(M, R + 64, generator=g) * 0.3).bfloat16()
    w = (torch.randn(N, R, generator=g) * 0.3).bfloat16()
    xin = torch.randn(M, N, generator=g)
    rng = torch.tensor([99, 3], dtype=torch.int64)
    for s, p, resid in ((1.0, 0.0, False), (0.25, 0.0, True), (1.0, 0.25, False), (2.0, 0.1, True)):
        yc, xo = y_full.clone(), torch.zeros(M, N)
        f.lora_apply(yc[:, 8:8 + N], u_full[:, 64:], w, M, N, R, s=s, p=p, rng=rng, sid=5, x_in=xin if resid else None, x_out=xo if resid else None)
        yg, xg = y_full.cuda(), torch.zeros(M, N, device="cuda")
        ops.lora_apply(yg[:, 8:8 + N], u_full.cuda()[:, 64:], w.cuda(), M, N, R, s=s, p=p, rng=rng.cuda(), sid=5,
                       x_in=xin.cuda() if resid else None, x_out=xg if resid else None)
        a, b = yg.cpu().float(), yc.float()
        assert torch.equal(a[:, :8], b[:, :8]) and torch.equal(a[:, 8 + N:], b[:, 8 + N:])
        diff = (a - b).abs()
        assert float((diff > 0).float().mean()) < 2e-3 and float(diff.max()) <= 2.0 ** -6 * float(b.abs().max())
        if p > 0:                                                   # the same elements are dropped
            base = y_full[:, 8:8 + N].float()
            assert torch.equal((a[:, 8:8 + N] == base), (b[:, 8:8 + N] == base)) or float(((a[:, 8:8 + N] == base) != (b[:, 8:8 + N] == base)).float().mean()) < 1e-4
        if resid:
            assert torch.equal(xg.cpu(), xin + a[:, 8:8 + N])


@pytest.mark.parametrize("name", ["mid_text_lora", "mid_text_lora_qv"])
def test_lora_refresh_one_launch_equals_the_copies(ops, name):
    """tasu_lora_refresh (all working copies of all adapters from the bucket's bf16 image in one launch) against the copies
    built one by one: s A, s B^T, A^T (zero-padded to the 64-wide rank), and B inside the K-extended weights [W | B]."""
    z, geo, cfg, sd, lsd, batch = golden_case(name)
    gm = build(geo, cfg, sd, lsd, ops, "cuda")
    lp, pb, sc = gm.lora, gm.proj.pb, cfg.scaling
    for l in range(geo.llm_layers):
        for t in cfg.target_modules:
            a, b = lp.view(pb, l, t, "A").float(), lp.view(pb, l, t, "B").float()
            assert torch.equal(lp.as_[(l, t)], (a * sc).bfloat16()), (l, t)
            assert torch.equal(lp.bts[(l, t)], (b.t() * sc).bfloat16()), (l, t)
            assert torch.equal(lp.at[(l, t)][:, : lp.r], a.t().bfloat16()) and (lp.rp == lp.r or float(lp.at[(l, t)][:, lp.r:].abs().max()) == 0)
            assert torch.equal(lp.b_ext(l, lp.group_of[t], t), b.bfloat16()), (l, t)
    # the K-extended weights: the frozen base weight in the head, every B in its member's rows / rank columns, zeros elsewhere
    for (l, g), we in lp.wext.items():
        K = lp.kbase[g]
        assert torch.equal(we[:, :K], gm.llm.layers[l][lp.wname[g]])
        tail = we[:, K:].clone()
        for t in dict(lp.groups)[g]:
            c0, o = lp.cols[t], lp.dims[t][1]
            k0 = lp.slot[t] * lp.rp
            tail[c0:c0 + o, k0:k0 + lp.r] = 0
        assert float(tail.abs().max()) == 0.0, (l, g)


def test_integration_md_lora_stub_runs_as_written():
    """The ctypes stub INTEGRATION.md shows for peft's lora.Linear.forward, executed verbatim (extracted from the document) on
    the C-ABI: result = base(x) + lora_B(lora_A(x)) * scaling against an fp32 product of the same bf16 operands."""
    import ctypes
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    base = next(b for b in blocks if "def linear(" in b)
    lora = next(b for b in blocks if "def lora_linear(" in b)
    ns = {}
    cwd = os.getcwd()
    os.chdir(root)                                              # the stub loads "ps_slm_amd/libtasu_hip.so" relative to the repo
    try:
        exec(base, ns)
        exec(lora, ns)
    finally:
        os.chdir(cwd)
    g = torch.Generator().manual_seed(0)
    M, K, N, r, s = 333, 256, 512, 64, 0.25
    x = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.1).bfloat16().cuda()
    a = (torch.randn(r, K, generator=g) * 0.1).bfloat16().cuda()
    b = (torch.randn(N, r, generator=g) * 0.1).bfloat16().cuda()
    y = ns["lora_linear"](x, w, a, b, s)
    torch.cuda.synchronize()
    ref = x.float() @ w.float().t() + (x.float() @ a.float().t()) @ b.float().t() * s
    assert float((y.float() - ref).abs().max()) < 2e-2 * float(ref.abs().max())


def test_lora_rank_above_64_hip_vs_double(ops):
    """r = 128: rank GEMMs on the tile policy (N > 64), two 64-wide rank blocks in the K-extended operands and in tasu_lora_apply."""
    from test_lora_cpu import rank_gt64_case
    geo, cfg, sd, lsd, batch = rank_gt64_case()
    gm, cm = build(geo, cfg, sd, lsd, ops, "cuda"), build(geo, cfg, sd, lsd, FakeOps(), "cpu")
    sg, sc = run_text(gm, batch), run_text(cm, batch)
    torch.cuda.synchronize()
    assert abs(float(sg.dev["loss_out"][0]) - float(sc.dev["loss_out"][0])) < 2e-3
    for (k, g1), (_, g2) in zip(sorted(gm.lora_grads().items()), sorted(cm.lora_grads().items())):
        assert cosine(g1, g2) > 0.9995, k
