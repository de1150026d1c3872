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
    t = torch.randn(n, generator=g).bfloat16()
    xin = torch.randn(n, generator=g)
    for s in (0.25, 2.0, 0.3):
        yc, xo = y.clone(), torch.empty(n)
        f.lora_add(yc, t, s, xin, xo)
        yg, xg = y.cuda(), torch.empty(n, device="cuda")
        ops.lora_add(yg, t.cuda(), s, xin.cuda(), xg)
        assert torch.equal(yg.cpu(), yc) and torch.equal(xg.cpu(), xo)
        yc2, yg2 = y.clone(), y.cuda()
        f.lora_add(yc2, t, s)
        ops.lora_add(yg2, t.cuda(), s)
        assert torch.equal(yg2.cpu(), yc2)
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
    ops.rng_advance(rng_g)
    assert rng_g.cpu().tolist() == [123456789, 5]
    # a different step or stream draws a different mask; the same (seed, step, stream) the same one
    a, b, c = (torch.empty(n, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    ones = torch.ones(n, dtype=torch.bfloat16, device="cuda")
    ops.lora_dropout(ones, a, 0.5, rng_g, 3)
    ops.lora_dropout(ones, b, 0.5, rng_g, 3)
    ops.lora_dropout(ones, c, 0.5, rng_g, 4)
    assert torch.equal(a, b) and not torch.equal(a, c)


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
