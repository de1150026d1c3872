"""GPU parity of every HIP kernel (through the C-ABI via ps_slm_amd.ops.HipOps) against the torch-CPU double in
tests/fake_ops.py on the same seeded inputs.  Integer / index results must be bit-exact; floating point
tolerances are stated per test (bf16 results: a few bf16 ulps = rel 2e-2 of the tensor scale; fp32: 1e-4)."""
import math

import numpy as np
import pytest
import torch

from fake_ops import FakeOps

pytestmark = pytest.mark.gpu
HD = 128
BF, F32, I32 = torch.bfloat16, torch.float32, torch.int32


@pytest.fixture(scope="module")
def hip():
    from ps_slm_amd.ops import HipOps
    return HipOps()


@pytest.fixture(params=["stream", "skinny"])
def hip_both(hip, request):
    """The decode-step GEMM ops on both kernel families: the single-launch streaming kernels (csrc/gemm_stream.hip, where they
    serve the shape) and the split-K + finish kernels (csrc/gemm_skinny.hip)."""
    hip.use_stream = hip.dec_down_slabs = request.param == "stream"     # K-range slabs for K = 8960 / the split-K kernels
    yield hip
    hip.use_stream, hip.dec_down_slabs = True, True


@pytest.fixture(scope="module")
def fake():
    return FakeOps()


def dev(t):
    return None if t is None else t.cuda()


def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def run_pair(hip, fake, name, args, outs):
    """args: list of CPU tensors / scalars; outs: indices of args that are outputs.  Returns (cpu, gpu) outs."""
    cargs = [a.clone() if isinstance(a, torch.Tensor) else a for a in args]
    gargs = [a.cuda() if isinstance(a, torch.Tensor) else a for a in args]
    getattr(fake, name)(*cargs)
    getattr(hip, name)(*gargs)
    torch.cuda.synchronize()
    return [cargs[i] for i in outs], [gargs[i].cpu() for i in outs]


def randn(*shape, dtype=F32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (200, 1000, 256), (130, 203, 192), (1, 64, 64),
                                   (1024, 1536, 1536), (512, 256, 8960)])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("bias", [False, True])
def test_gemm(hip, fake, M, N, K, mode, bias):
    ldc = (N + 63) // 64 * 64
    a = randn(M, K, dtype=BF, seed=1)
    b = randn(N, K, dtype=BF, seed=2, scale=1.0 / math.sqrt(K))
    bv = randn(N, dtype=BF, seed=3) if bias else None
    c = torch.zeros(M, ldc, dtype=BF if mode == 0 else F32)
    r = randn(M, ldc, seed=4) if mode == 2 else None
    cc, gc = c.clone(), c.cuda()
    fake.gemm(a, b, cc, M, N, K, bias=bv, resid=r, mode=mode)
    hip.gemm(a.cuda(), b.cuda(), gc, M, N, K, bias=dev(bv), resid=dev(r), mode=mode)
    torch.cuda.synchronize()
    tol = 1e-2 if mode != 1 else 2e-5 * math.sqrt(K)
    assert rel_err(gc, cc) < tol
    assert torch.equal(gc.cpu()[:, N:], cc[:, N:]), "columns beyond N must be untouched"


@pytest.mark.parametrize("M,N,K", [(512, 512, 256), (300, 520, 384), (1000, 1000, 1536), (257, 72, 256), (768, 2304, 1280)])
@pytest.mark.parametrize("mode,bias", [(0, False), (0, True), (1, True), (2, False)])
def test_gemm_pp256_kernel(hip, fake, M, N, K, mode, bias):
    """The 256 x 256 eight-wave kernel (csrc/gemm_pp.hip) called by name: against the fp32 double, and BIT FOR BIT against the
    loader-wave kernel of gemm_pipe.hip (both accumulate every output element over K in the same order).  Shapes with edge tiles
    in both directions (rows / columns past the matrix are clipped by the buffer descriptor, not per lane), 1 to 27 tiles per
    workgroup stream, 4 to 24 K-tiles."""
    ldc = (N + 63) // 64 * 64
    a = randn(M, K, dtype=BF, seed=1)
    b = randn(N, K, dtype=BF, seed=2, scale=1.0 / math.sqrt(K))
    bv = randn(N, dtype=BF, seed=3) if bias else None
    c = torch.zeros(M, ldc, dtype=BF if mode == 0 else F32)
    r = randn(M, ldc, seed=4) if mode == 2 else None
    cc, g_pp, g_pipe = c.clone(), c.cuda(), c.cuda()
    fake.gemm(a, b, cc, M, N, K, bias=bv, resid=r, mode=mode)
    ad, bd = a.cuda(), b.cuda()
    hip.gemm_on("pp256", ad, bd, g_pp, M, N, K, bias=dev(bv), resid=dev(r), mode=mode)
    hip.gemm_on("pipe128", ad, bd, g_pipe, M, N, K, bias=dev(bv), resid=dev(r), mode=mode)
    torch.cuda.synchronize()
    assert rel_err(g_pp, cc) < (1e-2 if mode != 1 else 2e-5 * math.sqrt(K))
    assert torch.equal(g_pp.cpu()[:, N:], cc[:, N:]), "columns beyond N must be untouched"
    assert torch.equal(g_pp, g_pipe)


def test_gemm_pp256_many_tiles_per_workgroup(hip):
    """4096 x 17920 x 1536 (gate|up of the benchmark: 1120 tiles, 4 or 5 per workgroup, the stream of K-tiles crosses tile
    boundaries) and 2048 x 9000 x 256 (4 K-tiles per tile: every K-tile pair stages into the next tile): bit-identical to the
    loader-wave kernel, run twice (bitwise repeatable: no race on the LDS ring)."""
    for M, N, K in ((4096, 17920, 1536), (2048, 9000, 256)):
        a = randn(M, K, dtype=BF, seed=5).cuda()
        b = randn(N, K, dtype=BF, seed=6, scale=1.0 / math.sqrt(K)).cuda()
        ldc = (N + 63) // 64 * 64
        c1, c2, c3 = (torch.zeros(M, ldc, dtype=BF).cuda() for _ in range(3))
        hip.gemm_on("pp256", a, b, c1, M, N, K)
        hip.gemm_on("pipe128", a, b, c2, M, N, K)
        hip.gemm_on("pp256", a, b, c3, M, N, K)
        torch.cuda.synchronize()
        assert torch.equal(c1, c2) and torch.equal(c1, c3)


@pytest.mark.parametrize("mode", [0, 2])
def test_gemm_policy_splits_the_columns_over_two_kernels(hip, mode):
    """4096 x 8960 (560 tiles of 256 x 256 = 2.19 rounds of 256 CUs) and 4096 x 17920: the dispatcher runs whole rounds on the
    256 x 256 kernel and the remaining columns on the loader-wave kernel in a second launch; every output element still sees
    its K range in the same order, so the result equals one kernel's, bit for bit."""
    for M, N, K in ((4096, 8960, 256), (4096, 17920, 384), (3000, 9100, 256)):
        a = randn(M, K, dtype=BF, seed=7).cuda()
        b = randn(N, K, dtype=BF, seed=8, scale=1.0 / math.sqrt(K)).cuda()
        ldc = (N + 63) // 64 * 64
        r = randn(M, ldc, seed=9).cuda() if mode == 2 else None
        c1 = torch.zeros(M, ldc, dtype=BF if mode == 0 else F32).cuda()
        c2 = torch.zeros_like(c1)
        hip.gemm(a, b, c1, M, N, K, resid=r, mode=mode)
        hip.gemm_on("pipe128", a, b, c2, M, N, K, resid=r, mode=mode)
        torch.cuda.synchronize()
        assert torch.equal(c1, c2), (M, N, K)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_gemm_unaligned_output_rows(hip, fake, mode):
    """ldc = N = 203: rows of C are neither 16- nor 8-byte aligned, so the kernels take their element-wise epilogues."""
    M, N, K = 300, 203, 128
    a = randn(M, K, dtype=BF, seed=1)
    b = randn(N, K, dtype=BF, seed=2, scale=1.0 / math.sqrt(K))
    bv = randn(N, dtype=BF, seed=3)
    c = torch.zeros(M, N, dtype=BF if mode == 0 else F32)
    r = randn(M, N, seed=4) if mode == 2 else None
    cc, gc = c.clone(), c.cuda()
    fake.gemm(a, b, cc, M, N, K, bias=bv, resid=r, mode=mode)
    hip.gemm(a.cuda(), b.cuda(), gc, M, N, K, bias=dev(bv), resid=dev(r), mode=mode)
    torch.cuda.synchronize()
    assert rel_err(gc, cc) < (1e-2 if mode != 1 else 2e-5 * math.sqrt(K))


@pytest.mark.parametrize("M,N,K", [(512, 384, 32768), (300, 200, 24576), (1664, 2048, 25088), (1024, 1536, 28672)])
@pytest.mark.parametrize("mode", [0, 2])
def test_gemm_split_k(hip, fake, M, N, K, mode):
    """Deep, small-grid problems take the 256 x 192 split-K path (last-arriver reduction through the workspace): same
    result as the double, bitwise repeatable, and the arrival counters are left zero."""
    ldc = (N + 63) // 64 * 64
    a = randn(M, K, dtype=BF, seed=1)
    b = randn(N, K, dtype=BF, seed=2, scale=1.0 / math.sqrt(K))
    c = torch.zeros(M, ldc, dtype=BF if mode == 0 else F32)
    r = randn(M, ldc, seed=4) if mode == 2 else None
    cc, g1, g2 = c.clone(), c.cuda(), c.cuda()
    fake.gemm(a, b, cc, M, N, K, resid=r, mode=mode)
    ad, bd, rd = a.cuda(), b.cuda(), dev(r)
    hip.gemm(ad, bd, g1, M, N, K, resid=rd, mode=mode)
    hip.gemm(ad, bd, g2, M, N, K, resid=rd, mode=mode)
    torch.cuda.synchronize()
    assert rel_err(g1, cc) < 1e-2
    assert torch.equal(g1, g2)
    assert int(hip.gemm_ws[:4096 * 4].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize("M,N,K", [(4096, 1536, 8960), (2048, 1536, 17920), (3000, 1500, 8192), (4096, 4608, 1024), (1000, 1000, 4096),
                                   (2304, 1280, 6144)])
@pytest.mark.parametrize("mode,bias", [(0, False), (1, True), (2, False)])
def test_gemm_streamk(hip, fake, M, N, K, mode, bias):
    """The stream-K schedule of the 256 x 256 kernel (tasu_gemm_nt_bf16_streamk): 96 / 48 / 72 tiles on 256 workgroups (every
    tile cut into 2 to 6 K ranges, edge tiles in both directions), 288 tiles (one whole round + 32 tiles: the last 288 tiles cut
    into 9-pair ranges), 16 tiles of 32 pairs (ranges too short: whole tiles) and 45 tiles of 48 pairs (8-pair ranges, a
    boundary within 4 pairs of a tile end snaps to it).  Against the fp32 double; BITWISE repeatable (the partial tiles are added in K order,
    whoever finishes first); the flag words of the workspace are left at zero; columns beyond N untouched."""
    ldc = (N + 63) // 64 * 64
    a = randn(M, K, dtype=BF, seed=1)
    b = randn(N, K, dtype=BF, seed=2, scale=1.0 / math.sqrt(K))
    bv = randn(N, dtype=BF, seed=3) if bias else None
    c = torch.zeros(M, ldc, dtype=BF if mode == 0 else F32)
    r = randn(M, ldc, seed=4) if mode == 2 else None
    cc, g1, g2 = c.clone(), c.cuda(), c.cuda()
    fake.gemm(a, b, cc, M, N, K, bias=bv, resid=r, mode=mode)
    ad, bd = a.cuda(), b.cuda()
    hip.gemm_streamk(ad, bd, g1, M, N, K, bias=dev(bv), resid=dev(r), mode=mode)
    hip.gemm_streamk(ad, bd, g2, M, N, K, bias=dev(bv), resid=dev(r), mode=mode)
    torch.cuda.synchronize()
    assert rel_err(g1, cc) < (1e-2 if mode != 1 else 2e-5 * math.sqrt(K))
    assert torch.equal(g1, g2)
    assert torch.equal(g1.cpu()[:, N:], cc[:, N:]), "columns beyond N must be untouched"
    assert int(hip.gemm_ws[:4096 * 4].view(torch.int32).abs().sum()) == 0


def test_gemm_policy_takes_streamk_for_d_gate_up(hip):
    """4096 x 1536 x 17920 (the MLP's input gradient: 96 tiles of 256 x 256 on 256 CUs behind 280 K-tiles): the dispatcher's
    choice is the stream-K schedule -- its result is the named entry point's, bit for bit, and NOT the one-round 128 x 192
    grid's (a different association of the fp32 sums), which it matches to fp32 rounding."""
    M, N, K = 4096, 1536, 17920
    a = randn(M, K, dtype=BF, seed=5).cuda()
    b = randn(N, K, dtype=BF, seed=6, scale=1.0 / math.sqrt(K)).cuda()
    c1, c2, c3 = (torch.zeros(M, N, dtype=F32).cuda() for _ in range(3))
    hip.gemm(a, b, c1, M, N, K, mode=1)
    hip.gemm_streamk(a, b, c2, M, N, K, mode=1)
    hip.gemm_on("pipe192", a, b, c3, M, N, K, mode=1)
    torch.cuda.synchronize()
    assert torch.equal(c1, c2)
    assert not torch.equal(c1, c3) and rel_err(c1, c3) < 1e-5


@pytest.mark.parametrize("M,I,K", [(4096, 8960, 1536), (2048, 8960, 1536), (300, 200, 128), (1100, 4480, 384), (1000, 1000, 256),
                                   (64, 96, 256), (4096, 1024, 256)])
def test_gemm_dswiglu(hip, fake, M, I, K):
    """Down projection's input gradient + SwiGLU backward behind one entry point (tasu_gemm_dswiglu: the GEMM into the dact
    scratch + tasu_swiglu_bwd; the form with the SwiGLU backward in the GEMM epilogue lives in the lab build only) -- the same
    bits as the GEMM with bf16 output followed by tasu_swiglu_bwd, on the 256 x 256 kernel + column tail (4096 / 2048 x 8960),
    the loader-wave tiles with edge tiles in both directions, and 64 rows."""
    dy = randn(M, K, dtype=BF, seed=1).cuda()
    wd_t = randn(I, K, dtype=BF, seed=2, scale=1.0 / math.sqrt(K)).cuda()
    gu = randn(M, 2 * I, dtype=BF, seed=3).cuda()
    dact = torch.zeros(M, I, dtype=BF).cuda()
    want = torch.zeros(M, 2 * I, dtype=BF).cuda()
    hip.gemm(dy, wd_t, dact, M, I, K)
    hip.swiglu_bwd(dact, gu, want, M, I)
    got = torch.full((M, 2 * I), 7.0, dtype=BF).cuda()
    hip.gemm_dswiglu(dy, wd_t, gu, got, torch.zeros(M, I, dtype=BF).cuda(), M, I, K)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    cc = torch.zeros(M, 2 * I, dtype=BF)
    fake.gemm_dswiglu(dy.cpu(), wd_t.cpu(), gu.cpu(), cc, None, M, I, K)
    assert rel_err(want, cc) < 2e-2


@pytest.mark.parametrize("M,H,G,K,bias", [(4096, 12, 2, 1536, True), (1000, 4, 2, 256, True), (300, 2, 1, 128, False), (4096, 28, 4, 3584, True),
                                          (64, 12, 2, 1536, True)])
def test_gemm_qkv_rope(hip, fake, M, H, G, K, bias):
    """q|k|v projection + bias + RoPE of the q and k heads in the GEMM's epilogue (tasu_gemm_qkv_rope): the same bits as the GEMM
    with bf16 output followed by tasu_rope_fwd -- 1.5B and 7B head counts, a last row tile of 232 / 44 / 64 rows, no bias; the
    v heads come out unrotated."""
    N = (H + 2 * G) * HD
    a = randn(M, K, dtype=BF, seed=1).cuda()
    w = randn(N, K, dtype=BF, seed=2, scale=1.0 / math.sqrt(K)).cuda()
    bv = randn(N, dtype=BF, seed=3).cuda() if bias else None
    ang = randn(M, 64, seed=4)
    cos, sin = torch.cos(ang).cuda(), torch.sin(ang).cuda()
    want = torch.zeros(M, N, dtype=BF).cuda()
    hip.gemm(a, w, want, M, N, K, bias=bv)
    plain = want.clone()
    hip.rope_fwd(want, cos, sin, None, None, None, 1, M, H, G)
    got = torch.full((M, N), 7.0, dtype=BF).cuda()
    hip.gemm_qkv_rope(a, w, bv, got, cos, sin, M, H, G, K)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert torch.equal(want[:, (H + G) * HD:], plain[:, (H + G) * HD:]) and not torch.equal(want[:, :HD], plain[:, :HD])
    cc = torch.zeros(M, N, dtype=BF)
    fake.gemm_qkv_rope(a.cpu(), w.cpu(), None if bv is None else bv.cpu(), cc, cos.cpu(), sin.cpu(), M, H, G, K)
    assert rel_err(want, cc) < 2e-2


def test_gemm_gate_up_swiglu_streamk(hip, fake):
    """gate|up + SwiGLU on 96 tiles behind K = 16384 (2048 rows, I = 1536): with the workspace the policy cuts the tiles along K;
    gate|up and the activation agree with the unfused double, bitwise repeatable."""
    M, I, K = 2048, 1536, 16384
    a = randn(M, K, dtype=BF, seed=1)
    w = randn(2 * I, K, dtype=BF, seed=2, scale=1.0 / math.sqrt(K))
    gu_c, act_c = torch.zeros(M, 2 * I, dtype=BF), torch.zeros(M, I, dtype=BF)
    fake.gemm_gate_up_swiglu(a, w, gu_c, act_c, M, I, K)
    ad, wd = a.cuda(), w.cuda()
    outs = []
    for _ in range(2):
        gu, act = torch.zeros(M, 2 * I, dtype=BF).cuda(), torch.zeros(M, I, dtype=BF).cuda()
        hip.gemm_gate_up_swiglu(ad, wd, gu, act, M, I, K)
        outs.append((gu, act))
    torch.cuda.synchronize()
    assert rel_err(outs[0][0], gu_c) < 1e-2 and rel_err(outs[0][1], act_c) < 2e-2
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert int(hip.gemm_ws[:4096 * 4].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize("M,I,K", [(4096, 8960, 1536), (300, 200, 128), (257, 72, 64), (64, 96, 256), (4096, 8192, 256), (1100, 4480, 384)])
def test_gemm_gate_up_swiglu(hip, fake, M, I, K):
    """Fused epilogue == GEMM followed by swiglu_fwd: same gate|up bits, same activation bits."""
    a = randn(M, K, dtype=BF, seed=1).cuda()
    w = randn(2 * I, K, dtype=BF, seed=2, scale=1.0 / math.sqrt(K)).cuda()
    gu1, act1 = torch.zeros(M, 2 * I, dtype=BF).cuda(), torch.zeros(M, I, dtype=BF).cuda()
    gu2, act2 = torch.zeros(M, 2 * I, dtype=BF).cuda(), torch.zeros(M, I, dtype=BF).cuda()
    hip.gemm_gate_up_swiglu(a, w, gu1, act1, M, I, K)
    hip.gemm(a, w, gu2, M, 2 * I, K)
    hip.swiglu_fwd(gu2, act2, M, I)
    torch.cuda.synchronize()
    ref = a.float() @ w.float().t()
    assert rel_err(gu1, ref) < 1e-2
    if M > 128:                        # same accumulation order as the plain GEMM only where that uses the same kernel
        assert torch.equal(gu1, gu2) and torch.equal(act1, act2)
    else:
        assert rel_err(gu1, gu2.cpu()) < 1e-2 and rel_err(act1, act2.cpu()) < 2e-2


def test_gemm_exact_integers(hip):
    """A = I (padded), asymmetric B: catches row/col swaps and k-permutation errors exactly."""
    M = N = K = 128
    a = torch.eye(M, K).to(BF)
    b = (torch.arange(N)[:, None] * 3 + torch.arange(K)[None, :] % 7).float().to(BF)   # small ints: exact in bf16?
    b = (torch.arange(N)[:, None] % 16 + (torch.arange(K)[None, :] % 5) * 16).to(BF)
    c = torch.zeros(M, N, dtype=F32).cuda()
    hip.gemm(a.cuda(), b.cuda(), c, M, N, K, mode=1)
    torch.cuda.synchronize()
    assert torch.equal(c.cpu(), (a.float() @ b.float().t()))


def test_gemm_rejects_bad_k(hip):
    from ps_slm_amd.ops import TasuOpError
    a = torch.zeros(64, 96, dtype=BF).cuda()
    with pytest.raises(TasuOpError):
        hip.gemm(a, a, torch.zeros(64, 64, dtype=BF).cuda(), 64, 64, 96)


def test_transpose_cast(hip, fake):
    src = randn(100, 203, dtype=BF, seed=5)
    dst = torch.full((256, 128), 7.0, dtype=BF)
    (c,), (g,) = run_pair(hip, fake, "transpose", [src, dst, 100, 203, 128, 256], [1])
    assert torch.equal(c, g)
    # 16-byte aligned leading dimensions: interior tiles take the vectorised path, edge tiles the element-wise one
    src = randn(200, 328, dtype=BF, seed=7)
    dst = torch.full((384, 256), 7.0, dtype=BF)
    (c,), (g,) = run_pair(hip, fake, "transpose", [src, dst, 200, 328, 256, 384], [1])
    assert torch.equal(c, g)
    x = randn(1000 * 37 + 3, seed=6)
    (c,), (g,) = run_pair(hip, fake, "cast_bf16", [x, torch.zeros_like(x, dtype=BF)], [1])
    assert torch.equal(c, g)


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("M,D", [(37, 256), (512, 1536), (70, 3584), (33, 320)])
def test_rmsnorm(hip, fake, M, D):
    x, w = randn(M, D, seed=1), 1 + 0.1 * randn(D, seed=2)
    (yc, rc), (yg, rg) = run_pair(hip, fake, "rmsnorm_fwd", [x, w, torch.zeros(M, D, dtype=BF), torch.zeros(M), 1e-6],
                                  [2, 3])
    assert rel_err(yg, yc) < 1e-2 and rel_err(rg, rc) < 1e-5
    dy = randn(M, D, dtype=BF, seed=3)
    for acc in (False, True):
        dx0 = randn(M, D, seed=4)
        (dc, bc), (dg, bg) = run_pair(hip, fake, "rmsnorm_bwd", [dy, x, w, rc, dx0, torch.zeros(M, D, dtype=BF), acc],
                                      [4, 5])
        assert rel_err(dg, dc) < 1e-4 and rel_err(bg, bc) < 1e-2


@pytest.mark.parametrize("R,D,Dp,out_dtype", [(20, 203, 256, BF), (8, 25055, 25088, BF), (16, 560, 560, F32), (70, 512, 512, BF), (9, 256, 320, BF)])
def test_layernorm(hip, fake, R, D, Dp, out_dtype):
    x = torch.zeros(R, Dp)
    x[:, :D] = randn(R, D, seed=1).abs() * 0.01
    x[torch.arange(R), torch.arange(R) % D] += 0.9           # posterior-like rows
    g, b = torch.zeros(Dp), torch.zeros(Dp)
    g[:D], b[:D] = 1 + 0.1 * randn(D, seed=2), 0.1 * randn(D, seed=3)
    outs = run_pair(hip, fake, "layernorm_fwd", [x, g, b, torch.zeros(R, Dp, dtype=out_dtype), torch.zeros(R), torch.zeros(R),
                                                 R, D, 1e-5], [3, 4, 5])
    (yc, mc, rc), (yg, mg, rg) = outs
    assert rel_err(yg, yc) < (1e-2 if out_dtype == BF else 1e-4)
    assert rel_err(mg, mc) < 1e-5 and rel_err(rg, rc) < 1e-4
    assert float(yg[:, D:].abs().max() if Dp > D else 0) == 0
    dy = randn(R, Dp, dtype=BF, seed=4)
    ws = torch.zeros(2 * 16 * D)
    (gc, bc), (gg, bg) = run_pair(hip, fake, "layernorm_bwd_params", [dy, x, mc, rc, torch.zeros(Dp), torch.zeros(Dp), ws, R, D],
                                  [4, 5])
    assert rel_err(gg, gc) < 1e-4 and rel_err(bg, bc) < 1e-4


def test_colsum(hip, fake):
    x = randn(333, 200, dtype=BF, seed=1)
    (c,), (g,) = run_pair(hip, fake, "colsum", [x, torch.zeros(200), 333, 200], [1])
    assert rel_err(g, c) < 1e-5


# ------------------------------------------------------------------------------------------------ rope + attention
def make_mask(B, S, kind):
    Spad = (S + 63) // 64 * 64
    m = torch.zeros(B, Spad, dtype=torch.uint8)
    for b in range(B):
        n = S - (b * 7) % max(S // 2, 1)
        if kind == "right":
            m[b, :n] = 1
        elif kind == "left":
            m[b, S - n:S] = 1
        else:
            m[b, :S] = 1
    return m


@pytest.mark.parametrize("B,S,H,G", [(2, 100, 4, 2), (2, 256, 12, 2), (1, 64, 2, 1), (1, 70, 28, 4)])
def test_rope(hip, fake, B, S, H, G):
    M, LD, Spad = B * S, (H + 2 * G) * HD, (S + 63) // 64 * 64
    pos = torch.randint(0, S, (M,), dtype=I32)
    (cc, sc), (cg, sg) = run_pair(hip, fake, "rope_table", [pos, torch.zeros(M, 64), torch.zeros(M, 64), HD, 1e6], [1, 2])
    assert rel_err(cg, cc) < 1e-5 and rel_err(sg, sc) < 1e-5
    qkv = randn(M, LD, dtype=BF, seed=1)
    outs = run_pair(hip, fake, "rope_fwd", [qkv, cc, sc, torch.ones(B * H * HD * Spad, dtype=BF), torch.ones(B * G * HD * Spad, dtype=BF),
                                            torch.ones(B * G * HD * Spad, dtype=BF), B, S, H, G], [0, 3, 4, 5])
    for c, g in zip(*outs):
        assert rel_err(g, c) < 1e-2
    dqkv = randn(M, LD, dtype=BF, seed=2)
    rep = H // G
    hpb = 3 if rep % 3 == 0 else (2 if rep % 2 == 0 else 1)
    dkp, dvp = randn(M, (H // hpb) * HD, seed=3), randn(M, (H // hpb) * HD, seed=4)
    (c,), (g,) = run_pair(hip, fake, "rope_bwd", [dqkv, dkp, dvp, cc, sc, B, S, H, G], [0])
    assert rel_err(g, c) < 1e-2


@pytest.mark.parametrize("B,S,H,G", [(2, 100, 4, 2), (2, 256, 12, 2), (3, 64, 2, 1), (1, 192, 4, 4)])
@pytest.mark.parametrize("mask_kind", ["right", "left", "none"])
@pytest.mark.parametrize("causal", [True, False])
def test_attention_fwd_bwd(hip, fake, B, S, H, G, mask_kind, causal):
    M, LD, Spad = B * S, (H + 2 * G) * HD, (S + 63) // 64 * 64
    scale = HD ** -0.5
    qkv = randn(M, LD, dtype=BF, seed=1)
    km = make_mask(B, S, mask_kind)
    cos, sin = torch.ones(M, 64), torch.zeros(M, 64)          # identity rotation: only the transposes matter here
    qt, kt, vt = (torch.zeros(B * n * HD * Spad, dtype=BF) for n in (H, G, G))
    fake.rope_fwd(qkv, cos, sin, qt, kt, vt, B, S, H, G)
    out, lse = torch.zeros(M, H * HD, dtype=BF), torch.zeros(B * H * Spad)
    (oc, lc), (og, lg) = run_pair(hip, fake, "attn_fwd", [qkv, vt, km, out, lse, B, S, H, G, scale, causal], [3, 4])
    live = km[:, :S].bool()                                   # query rows that are real tokens
    oc4, og4 = oc.view(B, S, H, HD), og.view(B, S, H, HD)
    assert rel_err(og4[live], oc4[live]) < 2e-2
    lcv, lgv = lc.view(B, H, Spad)[..., :S], lg.view(B, H, Spad)[..., :S]
    lm = live[:, None, :].expand(B, H, S)
    assert float((lgv - lcv)[lm].abs().max()) < 2e-3
    # backward
    dout = randn(M, H * HD, dtype=BF, seed=2)
    dout.view(B, S, H, HD)[~live] = 0                          # padded rows carry no gradient in the real model
    (dc, tc), (dg, tg) = run_pair(hip, fake, "attn_bwd_prep", [dout, oc, torch.zeros(B * H * Spad), torch.ones(B * H * HD * Spad, dtype=BF),
                                                               B, S, H], [2, 3])
    assert float((dg.view(B, H, Spad)[..., :S] - dc.view(B, H, Spad)[..., :S]).abs().max()) < 1e-3 * max(1.0, float(dc.abs().max()))
    assert torch.equal(tg, tc)
    dqkv = torch.zeros(M, LD, dtype=BF)
    (qc,), (qg,) = run_pair(hip, fake, "attn_bwd_dq", [qkv, kt, km, dout, lc, dc, dqkv, B, S, H, G, scale, causal], [6])
    assert rel_err(qg.view(B, S, -1)[:, :, :H * HD], qc.view(B, S, -1)[:, :, :H * HD]) < 2e-2
    dkp, dvp = torch.zeros(M, H * HD), torch.zeros(M, H * HD)
    (kc, vc), (kg, vg) = run_pair(hip, fake, "attn_bwd_dkv", [qkv, qt, km, dout, tc, lc, dc, dkp, dvp, B, S, H, G, scale, causal],
                                  [7, 8])
    rep = H // G
    hpb = 3 if rep % 3 == 0 else (2 if rep % 2 == 0 else 1)
    n_used = M * (H // hpb) * HD
    assert rel_err(kg.reshape(-1)[:n_used], kc.reshape(-1)[:n_used]) < 2e-2
    assert rel_err(vg.reshape(-1)[:n_used], vc.reshape(-1)[:n_used]) < 2e-2
    # the single-launch form gives the same bits as the two launches
    q2, k2, v2 = torch.zeros(M, LD, dtype=BF).cuda(), torch.zeros(M, H * HD).cuda(), torch.zeros(M, H * HD).cuda()
    hip.attn_bwd(qkv.cuda(), qt.cuda(), kt.cuda(), km.cuda(), dout.cuda(), tc.cuda(), lc.cuda(), dc.cuda(), q2, k2, v2, B, S, H, G,
                 scale, causal)
    torch.cuda.synchronize()
    assert torch.equal(q2.cpu().view(B, S, -1)[:, :, :H * HD], qg.view(B, S, -1)[:, :, :H * HD])
    assert torch.equal(k2.cpu().reshape(-1)[:n_used], kg.reshape(-1)[:n_used])
    assert torch.equal(v2.cpu().reshape(-1)[:n_used], vg.reshape(-1)[:n_used])


@pytest.mark.parametrize("B,S,H,G", [(2, 100, 4, 2), (2, 256, 12, 2), (3, 64, 2, 1), (1, 70, 28, 4), (2, 300, 12, 2), (1, 628, 12, 2),
                                     (2, 130, 10, 2), (16, 256, 12, 2), (16, 256, 28, 4), (1, 19, 6, 2)])
@pytest.mark.parametrize("mask_kind", ["right", "left", "none"])
@pytest.mark.parametrize("causal", [True, False])
def test_attention_bwd_gqa_kernel_equals_per_head_kernels(hip, fake, B, S, H, G, mask_kind, causal):
    """csrc/attention_gqa.hip (one launch: the query heads of a GQA group share the staged tiles, LDS-DMA ring, dK / dV complete
    per workgroup, RoPE backward in the epilogues) against the per-head kernels of rounds 1-3 + tasu_rope_bwd: the SAME BITS for
    the rotated dq; dk / dv equal up to the association of their fp32 sums -- for 2, 3, 5, 6 and 7 heads per group, ragged and
    padded lengths (S not a multiple of 64, more tiles than ring slots), causal and bidirectional; and the policy entry point
    agrees with whichever kernel it takes."""
    if B == 16 and (mask_kind != "none" or not causal):
        pytest.skip("the benchmark shapes once")
    M, LD, Spad = B * S, (H + 2 * G) * HD, (S + 63) // 64 * 64
    scale = HD ** -0.5
    assert hip.lib.tasu_attn_gqa_supported(S, H, G) == 1
    qkv = randn(M, LD, dtype=BF, seed=11).cuda()
    km = make_mask(B, S, mask_kind).cuda()
    pos = torch.arange(S, dtype=I32).repeat(B)
    cos, sin = torch.zeros(M, 64).cuda(), torch.zeros(M, 64).cuda()
    hip.rope_table(pos.cuda(), cos, sin, HD, 1e6)
    out, lse = torch.zeros(M, H * HD, dtype=BF).cuda(), torch.zeros(B * H * Spad).cuda()
    hip.attn_fwd(qkv, None, km, out, lse, B, S, H, G, scale, causal)
    live = km[:, :S].bool()
    dout = randn(M, H * HD, dtype=BF, seed=12).cuda()
    dout.view(B, S, H * HD)[~live] = 0
    delta = torch.zeros(B * H * Spad).cuda()
    hip.attn_bwd_prep(dout, out, delta, None, B, S, H)
    res = []
    for kernel in (1, 2, 0):                                    # TASU_ATTN_KERNEL_PER_HEAD, _GQA, _POLICY
        dqkv = torch.full((M, LD), 7.0, dtype=BF).cuda()
        dkp, dvp = torch.zeros(M, H * HD).cuda(), torch.zeros(M, H * HD).cuda()
        hip.attn_bwd_rope(qkv, km, dout, lse, delta, cos, sin, dqkv, dkp, dvp, B, S, H, G, scale, causal, kernel)
        res.append(dqkv)
    torch.cuda.synchronize()
    r1, r2, r0 = (r.view(B, S, LD)[live] for r in res)
    assert torch.isfinite(r1.float()).all()
    assert torch.equal(r1[:, :H * HD], r2[:, :H * HD])                              # dq: the same bits
    # dK / dV: the GQA workgroup sums ALL heads of the group (two query halves of every tile in two wave groups, one fp32 add at
    # the end) where the per-head kernels round-trip 1-7 fp32 partials through memory: the same products in another fp32
    # association -- equal up to bf16 roundings that flip (a 2^-9 step on a few elements in a thousand)
    for lo, hi in ((H * HD, (H + G) * HD), ((H + G) * HD, (H + 2 * G) * HD)):
        a, c = r1[:, lo:hi].float(), r2[:, lo:hi].float()
        assert rel_err(c, a) < 2 ** -7                                              # at most one bf16 step of the largest element
        assert float(((a - c).abs() > 2 ** -7 * a.abs().clamp_min(1e-6)).float().mean()) < 1e-3
    assert torch.equal(r0, r2)                                  # the policy: the GQA kernel wherever it is served


@pytest.mark.parametrize("B,S,H,G", [(2, 100, 4, 2), (1, 19, 6, 2), (2, 130, 10, 2), (1, 70, 28, 4)])
@pytest.mark.parametrize("causal", [True, False])
def test_attention_bwd_ignores_stale_bits_in_the_padding_rows(hip, fake, B, S, H, G, causal):
    """lse / delta are [B, H, Spad] buffers out of an uninitialised allocation (model.py ``_buf`` = torch.empty) and
    only rows < S are ever written by the forward / tasu_attn_bwd_prep.  With NaN in every word beforehand (what an int -100
    label word reads as in fp32), S % 64 != 0: the per-head kernels, the GQA kernel and the single-pass kernels must give finite
    gradients, bit-identical to the run on zero-filled buffers."""
    M, LD, Spad = B * S, (H + 2 * G) * HD, (S + 63) // 64 * 64
    scale = HD ** -0.5
    qkv = randn(M, LD, dtype=BF, seed=21).cuda()
    km = make_mask(B, S, "right").cuda()
    cos, sin = torch.zeros(M, 64).cuda(), torch.zeros(M, 64).cuda()
    hip.rope_table(torch.arange(S, dtype=I32).repeat(B).cuda(), cos, sin, HD, 1e6)
    dout = randn(M, H * HD, dtype=BF, seed=22).cuda()
    dout.view(B, S, H * HD)[~km[:, :S].bool()] = 0
    res = {}
    for fill in (0.0, float("nan")):
        out, lse = torch.zeros(M, H * HD, dtype=BF).cuda(), torch.full((B * H * Spad,), fill).cuda()
        hip.attn_fwd(qkv, None, km, out, lse, B, S, H, G, scale, causal)
        delta = torch.full((B * H * Spad,), fill).cuda()
        hip.attn_bwd_prep(dout, out, delta, None, B, S, H)
        assert torch.isfinite(delta).all()                      # the padding rows of delta are written (0), not skipped
        for kernel in (1, 2):                                   # TASU_ATTN_KERNEL_PER_HEAD, _GQA
            dqkv = torch.zeros(M, LD, dtype=BF).cuda()
            dkp, dvp = torch.zeros(M, H * HD).cuda(), torch.zeros(M, H * HD).cuda()
            delta_k = delta.clone()
            delta_k.view(B, H, Spad)[..., S:] = fill            # a caller that never ran prep over the padding (older ABI behaviour)
            hip.attn_bwd_rope(qkv, km, dout, lse, delta_k, cos, sin, dqkv, dkp, dvp, B, S, H, G, scale, causal, kernel)
            res[(fill == 0.0, kernel)] = dqkv
        if hip.lib.tasu_attn_sp_supported(S, H, G) == 1:
            dqkv = torch.zeros(M, LD, dtype=BF).cuda()
            dkp, dvp = torch.zeros(M, H * HD).cuda(), torch.zeros(M, H * HD).cuda()
            hip.attn_bwd_fused(qkv, km, dout, out, lse, torch.full((B * H * Spad,), fill).cuda(), cos, sin, dqkv, dkp, dvp, B, S, H, G,
                               scale, causal, "sp")
            res[(fill == 0.0, 3)] = dqkv
    torch.cuda.synchronize()
    for (clean, kernel), g in res.items():
        assert torch.isfinite(g.float()).all(), (clean, kernel)
        assert torch.equal(g, res[(True, kernel)]), kernel


@pytest.mark.parametrize("B,S,H,G", [(2, 100, 4, 2), (2, 256, 12, 2), (3, 64, 2, 1), (1, 192, 4, 4), (1, 19, 6, 2), (2, 130, 10, 2),
                                     (1, 256, 28, 4), (2, 249, 12, 2), (16, 256, 12, 2)])
@pytest.mark.parametrize("mask_kind", ["right", "left", "none"])
@pytest.mark.parametrize("causal", [True, False])
def test_attention_single_pass_kernels(hip, fake, B, S, H, G, mask_kind, causal):
    """csrc/attention_sp.hip (Spad <= 256: the whole K / V or Q / dO of a (batch, head) resident in LDS, one softmax, delta inside
    the backward kernel, per-query-head dK / dV partials + the reduce / rotary launch) against the torch double AND against the
    tiled kernels: forward out / lse, and the finished gradient of the unrotated q | k | v projection, for ragged lengths (S not
    a multiple of 16 / 32 / 64 / 128), 1-7 heads per group, left / right key padding, causal and bidirectional."""
    if B == 16 and (mask_kind != "none" or not causal):
        pytest.skip("the benchmark shape once")
    M, LD, Spad = B * S, (H + 2 * G) * HD, (S + 63) // 64 * 64
    scale = HD ** -0.5
    assert hip.lib.tasu_attn_sp_supported(S, H, G) == 1 and hip.lib.tasu_attn_sp_supported(257, H, G) == 0
    qkv = randn(M, LD, dtype=BF, seed=21)
    km = make_mask(B, S, mask_kind)
    live = km[:, :S].bool()
    pos = torch.arange(S, dtype=I32).repeat(B)
    cos, sin = torch.zeros(M, 64), torch.zeros(M, 64)
    fake.rope_table(pos, cos, sin, HD, 1e6)
    # forward: double, tiled kernel, single-pass kernel
    oc, lc = torch.zeros(M, H * HD, dtype=BF), torch.zeros(B * H * Spad)
    fake.attn_fwd(qkv, None, km, oc, lc, B, S, H, G, scale, causal)
    res = {}
    for kernel in ("tiled", "sp"):
        o, l = torch.full((M, H * HD), 3.0, dtype=BF).cuda(), torch.zeros(B * H * Spad).cuda()
        hip.attn_fwd_on(kernel, qkv.cuda(), km.cuda(), o, l, B, S, H, G, scale, causal)
        res[kernel] = (o.cpu(), l.cpu())
    lm = live[:, None, :].expand(B, H, S)
    for kernel in ("tiled", "sp"):
        o, l = res[kernel]
        assert rel_err(o.view(B, S, H, HD)[live], oc.view(B, S, H, HD)[live]) < 2e-2, kernel
        assert float((l.view(B, H, Spad)[..., :S] - lc.view(B, H, Spad)[..., :S])[lm].abs().max()) < 2e-3, kernel
    # the two kernels differ in where P is rounded to bf16 (against the running / the final row maximum) and in the association
    # of the sums: a few bf16 steps of the largest element
    a, c = res["sp"][0].view(B, S, H, HD)[live].float(), res["tiled"][0].view(B, S, H, HD)[live].float()
    assert rel_err(a, c) < 2 ** -6
    # the policy entry point: the tiled kernel (measured faster at every shape, tools/bench_attn_sp.py)
    o, l = torch.zeros(M, H * HD, dtype=BF).cuda(), torch.zeros(B * H * Spad).cuda()
    hip.attn_fwd(qkv.cuda(), None, km.cuda(), o, l, B, S, H, G, scale, causal)
    assert torch.equal(o.cpu().view(B, S, H, HD)[live], res["tiled"][0].view(B, S, H, HD)[live])
    # backward: the whole chain behind one entry point
    dout = randn(M, H * HD, dtype=BF, seed=22)
    dout.view(B, S, H, HD)[~live] = 0
    out_g, lse_g = res["sp"][0], res["sp"][1]
    want = torch.zeros(M, LD, dtype=BF)
    fake.attn_bwd_fused(qkv, km, dout, out_g, lse_g, torch.zeros(B * H * Spad), cos, sin, want, torch.zeros(M, H * HD),
                        torch.zeros(M, H * HD), B, S, H, G, scale, causal)
    got = {}
    for kernel in ("tiled", "sp", "policy"):
        dq = torch.full((M, LD), 7.0, dtype=BF).cuda()
        hip.attn_bwd_fused(qkv.cuda(), km.cuda(), dout.cuda(), out_g.cuda(), lse_g.cuda(), torch.zeros(B * H * Spad).cuda(), cos.cuda(),
                           sin.cuda(), dq, torch.zeros(M, H * HD).cuda(), torch.zeros(M, H * HD).cuda(), B, S, H, G, scale, causal, kernel)
        got[kernel] = dq.cpu().view(B, S, LD)[live]
    w = want.view(B, S, LD)[live]
    assert torch.isfinite(got["sp"].float()).all()
    for lo, hi in ((0, H * HD), (H * HD, (H + G) * HD), ((H + G) * HD, (H + 2 * G) * HD)):
        for kernel in ("tiled", "sp"):
            assert rel_err(got[kernel][:, lo:hi], w[:, lo:hi]) < 2e-2, (kernel, lo)
        a, c = got["sp"][:, lo:hi].float(), got["tiled"][:, lo:hi].float()
        assert rel_err(a, c) < 2 ** -6                     # the same products in another fp32 association
    # policy: single pass while its three roles fit one round of the chip, else tasu_attn_bwd_rope's policy (the GQA kernel where served)
    if 3 * B * H <= 320:
        assert torch.equal(got["policy"], got["sp"])
    else:
        gq = torch.full((M, LD), 7.0, dtype=BF).cuda()
        hip.attn_bwd_fused(qkv.cuda(), km.cuda(), dout.cuda(), out_g.cuda(), lse_g.cuda(), torch.zeros(B * H * Spad).cuda(), cos.cuda(),
                           sin.cuda(), gq, torch.zeros(M, H * HD).cuda(), torch.zeros(M, H * HD).cuda(), B, S, H, G, scale, causal,
                           "gqa" if H // G >= 2 else "tiled")
        assert torch.equal(got["policy"], gq.cpu().view(B, S, LD)[live])
    # bitwise repeatable (no atomics anywhere in the chain)
    dq2 = torch.zeros(M, LD, dtype=BF).cuda()
    hip.attn_bwd_fused(qkv.cuda(), km.cuda(), dout.cuda(), out_g.cuda(), lse_g.cuda(), None, cos.cuda(), sin.cuda(), dq2,
                       torch.zeros(M, H * HD).cuda(), torch.zeros(M, H * HD).cuda(), B, S, H, G, scale, causal, "sp")
    assert torch.equal(dq2.cpu().view(B, S, LD)[live], got["sp"])


def test_attention_online_softmax_rescale(hip, fake):
    """Force the running-max rescale branch: one late key dominates every earlier tile."""
    B, S, H, G = 1, 256, 2, 1
    M, LD, Spad = B * S, (H + 2 * G) * HD, 256
    qkv = randn(M, LD, dtype=BF, seed=3, scale=0.3)
    v = qkv.view(B, S, H + 2 * G, HD)
    v[0, 200, H] = v[0, 255, 0] * 8.0                           # key 200 spikes against query 255 (head 0)
    km = make_mask(B, S, "none")
    qt, kt, vt = (torch.zeros(B * n * HD * Spad, dtype=BF) for n in (H, G, G))
    fake.rope_fwd(qkv, torch.ones(M, 64), torch.zeros(M, 64), qt, kt, vt, B, S, H, G)
    for kernel in ("tiled", "sp"):                             # (the single-pass kernel has no rescale: the same answer)
        (oc, lc), (og, lg) = run_pair(hip, fake, "attn_fwd_on", [kernel, qkv, km, torch.zeros(M, H * HD, dtype=BF), torch.zeros(B * H * Spad),
                                                                 B, S, H, G, HD ** -0.5, True], [3, 4])
        assert rel_err(og, oc) < 2e-2 and float((lg - lc).abs().max()) < 2e-3


@pytest.mark.parametrize("D", [256, 1536, 3584, 512])
def test_rmsnorm_row_indexed(hip, fake, D):
    """Gathered forward / scattered backward over the labelled rows (tasu_rmsnorm_fwd_rows / _bwd_rows): the gathered rows must
    be BIT-identical to the plain kernels on the same rows (same arithmetic and summation order), unlabelled rows exact zeros."""
    M, n = 150, 64
    x = randn(M, D, seed=1)
    w = 1 + 0.1 * randn(D, seed=2)
    g = torch.Generator().manual_seed(3)
    rows = torch.randperm(M, generator=g)[:50].sort().values.to(I32)
    src = torch.full((n,), -1, dtype=I32)
    src[:50] = rows
    slot = torch.full((M,), -1, dtype=I32)
    slot[rows.long()] = torch.arange(50, dtype=I32)
    (yc, rc), (yg, rg) = run_pair(hip, fake, "rmsnorm_fwd_rows", [x, src, w, torch.ones(n, D, dtype=BF), torch.ones(n), 1e-6], [3, 4])
    assert rel_err(yg, yc) < 1e-2 and rel_err(rg, rc) < 1e-5
    assert float(yg[50:].float().abs().max()) == 0.0 and float(rg[50:].abs().max()) == 0.0
    y_full, r_full = torch.zeros(M, D, dtype=BF, device="cuda"), torch.zeros(M, device="cuda")
    hip.rmsnorm_fwd(x.cuda(), w.cuda(), y_full, r_full, 1e-6)
    torch.cuda.synchronize()
    assert torch.equal(y_full.cpu()[rows.long()], yg[:50]) and torch.equal(r_full.cpu()[rows.long()], rg[:50])
    dy = randn(n, D, dtype=BF, seed=4)
    (dc, dbc), (dg, dbg) = run_pair(hip, fake, "rmsnorm_bwd_rows", [dy, x, w, rc, slot, torch.ones(M, D), torch.ones(M, D, dtype=BF)], [5, 6])
    assert rel_err(dg, dc) < 1e-4 and rel_err(dbg, dbc) < 1e-2
    mask = torch.ones(M, dtype=torch.bool)
    mask[rows.long()] = False
    assert float(dg[mask].abs().max()) == 0.0 and float(dbg[mask].float().abs().max()) == 0.0
    dy_full = torch.zeros(M, D, dtype=BF)
    dy_full[rows.long()] = dy[:50]
    dx_full, dx_rows = torch.zeros(M, D, device="cuda"), torch.ones(M, D, device="cuda")
    hip.rmsnorm_bwd(dy_full.cuda(), x.cuda(), w.cuda(), r_full, dx_full, None, False)
    hip.rmsnorm_bwd_rows(dy.cuda(), x.cuda(), w.cuda(), rg.cuda(), slot.cuda(), dx_rows, None)      # the GPU's own rstd
    torch.cuda.synchronize()
    assert torch.equal(dx_full.cpu()[rows.long()], dx_rows.cpu()[rows.long()])


# ------------------------------------------------------------------------------------------------ activations
def test_swiglu_silu_relu(hip, fake):
    M, I = 77, 512
    gu = randn(M, 2 * I, dtype=BF, seed=1, scale=2.0)
    (c,), (g,) = run_pair(hip, fake, "swiglu_fwd", [gu, torch.zeros(M, I, dtype=BF), M, I], [1])
    assert rel_err(g, c) < 1e-2
    dact = randn(M, I, dtype=BF, seed=2)
    (c,), (g,) = run_pair(hip, fake, "swiglu_bwd", [dact, gu, torch.zeros(M, 2 * I, dtype=BF), M, I], [2])
    assert rel_err(g, c) < 1e-2
    x = randn(1003, dtype=BF, seed=3, scale=3.0)
    (c,), (g,) = run_pair(hip, fake, "silu_fwd", [x, torch.zeros_like(x)], [1])
    assert rel_err(g, c) < 1e-2
    (c,), (g,) = run_pair(hip, fake, "silu_bwd", [randn(1003, dtype=BF, seed=4), x, torch.zeros_like(x)], [2])
    assert rel_err(g, c) < 1e-2
    (c,), (g,) = run_pair(hip, fake, "relu_fwd", [x, torch.zeros_like(x)], [1])
    assert torch.equal(c, g)


# ------------------------------------------------------------------------------------------------ loss
@pytest.mark.parametrize("M,V,ldv", [(50, 1000, 1024), (6, 151936, 151936)])
def test_cross_entropy(hip, fake, M, V, ldv):
    lg = torch.zeros(M, ldv, dtype=BF)
    lg[:, :V] = randn(M, V, dtype=BF, seed=1, scale=3.0)
    lab = torch.randint(0, V, (M,), dtype=I32)
    lab[::3] = -100
    lab[1] = int(lg[1, :V].float().argmax())                   # at least one hit
    cnt = float((lab >= 0).sum())
    inv = torch.tensor([1.0 / cnt])
    args = [lg, lab, M, V, torch.zeros(M), torch.zeros(M, dtype=I32), torch.zeros(M, dtype=I32), torch.ones(M, ldv, dtype=BF), inv]
    (lc, hc, ac, dc), (lgp, hg, ag, dg) = run_pair(hip, fake, "ce_fwd_bwd", args, [4, 5, 6, 7])
    assert rel_err(lgp, lc) < 1e-4
    assert torch.equal(hg, hc) and torch.equal(ag, ac)
    assert rel_err(dg, dc) < 1e-2
    (oc,), (og,) = run_pair(hip, fake, "ce_reduce", [lc, hc, lab, M, torch.zeros(4)], [4])
    assert rel_err(og, oc) < 1e-5


def test_cross_entropy_in_place(hip, fake):
    """dlogits aliasing logits (what the training step does with keep_logits=False, ps_slm_amd/model.py): row_loss must still be
    lse - logit[label] -- the label's logit is read before any wave overwrites the row -- and the gradient must equal the
    two-buffer result.  4096 rows x 5 launches so that a lost race would show."""
    M, V, ldv = 4096, 1000, 1024
    lg = torch.zeros(M, ldv, dtype=BF)
    lg[:, :V] = randn(M, V, dtype=BF, seed=11, scale=3.0)
    lab = torch.randint(V - 64, V, (M,), dtype=I32)             # labels in the LAST chunk of the row: pass 2 reaches them late,
    lab[::2] = torch.randint(0, 64, (M // 2,), dtype=I32)        # and in the first chunk: pass 2 rewrites them first
    lab[::5] = -100
    inv = torch.tensor([1.0 / float((lab >= 0).sum())])
    want_l, want_h, want_d = torch.zeros(M), torch.zeros(M, dtype=I32), torch.zeros(M, ldv, dtype=BF)
    fake.ce_fwd_bwd(lg.clone(), lab, M, V, want_l, want_h, None, want_d, inv)
    for _ in range(5):
        buf, rl, rh = lg.cuda(), torch.zeros(M, device="cuda"), torch.zeros(M, dtype=I32, device="cuda")
        hip.ce_fwd_bwd(buf, lab.cuda(), M, V, rl, rh, None, buf, inv.cuda())
        torch.cuda.synchronize()
        assert float((rl.cpu() - want_l).abs().max()) < 1e-3 * float(want_l.abs().max())
        assert torch.equal(rh.cpu(), want_h)
        assert rel_err(buf, want_d) < 1e-2


def test_cross_entropy_row_in_registers_in_place(hip, fake):
    """Round 5: the training step's call (dlogits given, no argmax buffer, V >= 8192) runs on ce_reg_kernel -- the row's chunks
    stay in registers between the statistics and the gradient pass (one read of the logits).  Full vocabulary incl. the pad
    columns of the lm_head's Vpad, in place (dlogits aliasing logits), labels in the first and the last chunk, ignored rows:
    loss, hits and gradient against the double; twice the same bits."""
    M, V, ldv = 96, 151936, 151936
    lg = randn(M, ldv, dtype=BF, seed=21, scale=3.0)
    lab = torch.randint(0, V, (M,), dtype=I32)
    lab[::2] = torch.randint(V - 8, V, (M // 2,), dtype=I32)
    lab[1::4] = torch.randint(0, 8, (M // 4,), dtype=I32)
    lab[::5] = -100
    lab[3] = int(lg[3, :V].float().argmax())
    inv = torch.tensor([1.0 / float((lab >= 0).sum())])
    want_l, want_h, want_d = torch.zeros(M), torch.zeros(M, dtype=I32), torch.zeros(M, ldv, dtype=BF)
    fake.ce_fwd_bwd(lg.clone(), lab, M, V, want_l, want_h, None, want_d, inv)
    outs = []
    for _ in range(2):
        buf, rl, rh = lg.cuda(), torch.zeros(M, device="cuda"), torch.zeros(M, dtype=I32, device="cuda")
        hip.ce_fwd_bwd(buf, lab.cuda(), M, V, rl, rh, None, buf, inv.cuda())
        torch.cuda.synchronize()
        assert float((rl.cpu() - want_l).abs().max()) < 1e-4 * float(want_l.abs().max())
        assert torch.equal(rh.cpu(), want_h) and int(want_h.sum()) >= 1
        assert rel_err(buf, want_d) < 1e-2
        outs.append((buf.cpu(), rl.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # a narrower padded row (V not a multiple of 8 inside ldv): the pad columns get zero gradient
    V2, ld2 = 9001, 9088
    lg2 = torch.zeros(8, ld2, dtype=BF)
    lg2[:, :V2] = randn(8, V2, dtype=BF, seed=22, scale=2.0)
    lab2 = torch.randint(0, V2, (8,), dtype=I32)
    inv2 = torch.tensor([1.0 / 8])
    (lc, hc, dc), (lg_, hg, dg) = run_pair(hip, fake, "ce_fwd_bwd", [lg2, lab2, 8, V2, torch.zeros(8), torch.zeros(8, dtype=I32), None,
                                                                       torch.ones(8, ld2, dtype=BF), inv2], [4, 5, 7])
    assert rel_err(lg_, lc) < 1e-4 and torch.equal(hg, hc) and rel_err(dg, dc) < 1e-2 and float(dg[:, V2:].float().abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ front end
def test_posterior_merge_adamw(hip, fake):
    R, V, ld = 40, 203, 256
    ids = torch.randint(0, V, (R,), dtype=I32)
    ids[5::7] = -1
    alpha = torch.rand(R) * 0.1
    (c,), (g,) = run_pair(hip, fake, "posterior_build", [ids, alpha, torch.ones(R, ld), R, V], [2])
    assert torch.equal(c, g)
    M, D, Vt = 60, 256, 500
    table, proj = randn(Vt, D, seed=1), randn(R, D, dtype=BF, seed=2)
    kind = torch.randint(0, 3, (M,), dtype=I32)
    idx = torch.where(kind == 1, torch.randint(0, Vt, (M,)), torch.randint(0, R, (M,))).to(I32)
    (c,), (g,) = run_pair(hip, fake, "embed_merge", [table, proj, kind, idx, torch.ones(M, D), M, D], [4])
    assert torch.equal(c, g)
    rows = torch.randint(-1, M, (R,), dtype=I32)
    (c,), (g,) = run_pair(hip, fake, "merge_bwd", [randn(M, D, seed=3), rows, torch.ones(R, D, dtype=BF), R, D], [2])
    assert torch.equal(c, g)
    n = 4096 + 64
    p, gr, m, v = randn(n, seed=4), randn(n, seed=5), randn(n, seed=6).abs() * 0.1, randn(n, seed=7).abs() * 0.01
    lr = 5e-5
    outs = run_pair(hip, fake, "adamw", [p, gr, m, v, torch.zeros(n, dtype=BF), lr, 0.9, 0.999, 1e-6, 0.01, 3, 0.125], [0, 2, 3, 4])
    for c, g in zip(*outs):
        assert rel_err(g, c) < 1e-5 if c.dtype == F32 else rel_err(g, c) < 1e-2


# ------------------------------------------------------------------------------------------------ encoder / PSD
def test_encoder_aux(hip, fake):
    B, T, D = 2, 37, 80
    x = randn(B * T, D, seed=1)
    (c,), (g,) = run_pair(hip, fake, "sinusoid_pe", [x, torch.zeros(B * T, D), B, T, D, 16.0], [1])
    assert rel_err(g, c) < 1e-5
    E, ks = 256, 11
    qkv = randn(B * T, 3 * E, dtype=BF, seed=2)
    w = randn(E, ks, seed=3, scale=0.2)
    lens = torch.tensor([37, 20], dtype=I32)
    vcol = qkv[:, 2 * E:]
    c = torch.zeros(B * T, E)
    fake.fsmn_fwd(vcol, 3 * E, w, lens, c, B, T, E, ks, False)
    gq = qkv.cuda()
    g = torch.zeros(B * T, E).cuda()
    hip.fsmn_fwd(gq[:, 2 * E:], 3 * E, w.cuda(), lens.cuda(), g, B, T, E, ks, False)
    assert rel_err(g, c) < 1e-5
    fake.fsmn_fwd(vcol, 3 * E, w, lens, c, B, T, E, ks, True)              # accumulate on top (vector path)
    hip.fsmn_fwd(gq[:, 2 * E:], 3 * E, w.cuda(), lens.cuda(), g, B, T, E, ks, True)
    assert rel_err(g, c) < 1e-5
    E2 = 84                                                                 # not a multiple of 8: element-wise path
    q2 = randn(B * T, E2, dtype=BF, seed=5)
    w2 = randn(E2, ks, seed=6, scale=0.2)
    c2, g2 = torch.zeros(B * T, E2), torch.zeros(B * T, E2).cuda()
    fake.fsmn_fwd(q2, E2, w2, lens, c2, B, T, E2, ks, False)
    hip.fsmn_fwd(q2.cuda(), E2, w2.cuda(), lens.cuda(), g2, B, T, E2, ks, False)
    assert rel_err(g2, c2) < 1e-5
    R, V, ld = 30, 203, 256
    lg = randn(R, ld, seed=4, scale=3.0)
    (c,), (g,) = run_pair(hip, fake, "softmax_rows", [lg, torch.ones(R, ld), R, V], [1])
    assert rel_err(g, c) < 1e-5 and float(g[:, V:].abs().max()) == 0
    (c,), (g,) = run_pair(hip, fake, "softmax_rows", [lg.to(BF), torch.ones(R, ld), R, V], [1])
    assert rel_err(g, c) < 1e-5


def test_encoder_fsmn_and_ffn_relu_at_sensevoice_size(hip, fake):
    """The two encoder kernels of round 4 at SenseVoiceSmall's layer shape (512 channels, kernel 11, 504-frame rows, ragged):
    the row-blocked FSMN (4 frames x 8 channels per thread, taps in registers) against the CPU double, with and without
    accumulation; and PositionwiseFeedForward's w_1 + ReLU in the GEMM epilogue against GEMM + tasu_relu_fwd -- the same bits --
    on the kernels the encoder's M = 8064 takes and on the small-tile fallback (M = 100)."""
    B, T, E, ks = 3, 504, 512, 11
    qkv = randn(B * T, 3 * E, dtype=BF, seed=21)
    w = randn(E, ks, seed=22, scale=0.2)
    lens = torch.tensor([504, 333, 7], dtype=I32)
    c = randn(B * T, E, seed=23)
    g = c.clone().cuda()
    fake.fsmn_fwd(qkv[:, 2 * E:], 3 * E, w, lens, c, B, T, E, ks, True)
    gq = qkv.cuda()
    hip.fsmn_fwd(gq[:, 2 * E:], 3 * E, w.cuda(), lens.cuda(), g, B, T, E, ks, True)
    assert rel_err(g, c) < 1e-5
    c0, g0 = torch.ones(B * T, E), torch.ones(B * T, E).cuda()
    fake.fsmn_fwd(qkv[:, 2 * E:], 3 * E, w, lens, c0, B, T, E, ks, False)
    hip.fsmn_fwd(gq[:, 2 * E:], 3 * E, w.cuda(), lens.cuda(), g0, B, T, E, ks, False)
    assert rel_err(g0, c0) < 1e-5 and float(g0.view(B, T, E)[2, 7:].abs().max()) == 0       # frames past the length: exact zeros
    for M, N, K in ((8064, 2048, 512), (100, 2048, 512), (4096, 1024, 256)):
        a = randn(M, K, dtype=BF, seed=M).cuda()
        wt = (randn(N, K, seed=M + 1) * K ** -0.5).to(BF).cuda()
        bias = randn(N, dtype=BF, seed=M + 2).cuda()
        ref = torch.empty(M, N, dtype=BF, device="cuda")
        hip.gemm(a, wt, ref, M, N, K, bias=bias)
        hip.relu_fwd(ref, ref)
        out = torch.full((M, N), -1.0, dtype=BF, device="cuda")
        hip.gemm_bias_relu(a, wt, out, M, N, K, bias)
        torch.cuda.synchronize()
        assert torch.equal(out, ref) and float(out.float().min()) == 0.0 and float((out == 0).float().mean()) > 0.2


def test_fsmn_layernorm_one_launch_is_bit_identical(hip):
    """tasu_fsmn_ln_fwd (x += fsmn(v); xn = LayerNorm(x) in one launch: a wave owns 4 whole 512-channel rows) against
    tasu_fsmn_fwd + tasu_layernorm_fwd: the same bits in x and xn; ragged lengths, T not a multiple of the 16-frame block, a wider
    ldy with zero pad columns; and the fallback (another width) through the two kernels."""
    B, T, E, ks = 4, 501, 512, 11
    qkv = randn(B * T, 3 * E, dtype=BF, seed=31).cuda()
    w = randn(E, ks, seed=32, scale=0.2).cuda()
    lens = torch.tensor([501, 333, 7, 0], dtype=I32).cuda()
    x0 = randn(B * T, E, seed=33).cuda()
    ga, be = (1 + 0.1 * randn(E, seed=34)).cuda(), (0.1 * randn(E, seed=35)).cuda()
    x_a, x_b = x0.clone(), x0.clone()
    y_a = torch.full((B * T, E + 64), 3.0, dtype=BF, device="cuda")
    y_b = torch.full((B * T, E + 64), 3.0, dtype=BF, device="cuda")
    hip.fsmn_fwd(qkv[:, 2 * E:], 3 * E, w, lens, x_a, B, T, E, ks, True)
    hip.layernorm_fwd(x_a, ga, be, y_a, None, None, B * T, E, 1e-5)
    hip.fsmn_ln_fwd(qkv[:, 2 * E:], 3 * E, w, lens, x_b, ga, be, y_b, B, T, E, ks, 1e-5)
    torch.cuda.synchronize()
    assert torch.equal(x_a, x_b) and torch.equal(y_a, y_b) and float(y_b[:, E:].abs().max()) == 0.0
    assert not torch.equal(x_b, x0)
    # a width the fused kernel does not serve: the two kernels behind the same entry point
    E2 = 256
    v2 = randn(B * T, E2, dtype=BF, seed=36).cuda()
    w2, g2, b2 = randn(E2, ks, seed=37, scale=0.2).cuda(), torch.ones(E2).cuda(), torch.zeros(E2).cuda()
    xa, xb = randn(B * T, E2, seed=38).cuda(), None
    xb = xa.clone()
    ya, yb = torch.zeros(B * T, E2, dtype=BF, device="cuda"), torch.zeros(B * T, E2, dtype=BF, device="cuda")
    hip.fsmn_fwd(v2, E2, w2, lens, xa, B, T, E2, ks, True)
    hip.layernorm_fwd(xa, g2, b2, ya, None, None, B * T, E2, 1e-5)
    hip.fsmn_ln_fwd(v2, E2, w2, lens, xb, g2, b2, yb, B, T, E2, ks, 1e-5)
    torch.cuda.synchronize()
    assert torch.equal(xa, xb) and torch.equal(ya, yb)


def test_psd_kernels(hip, fake):
    from conftest import load_npz
    z = load_npz("psd_crafted")
    post = torch.from_numpy(z["posterior"])                     # [B, T, V]
    B, T, V = post.shape
    ld = 64
    p2 = torch.zeros(B * T, ld)
    p2[:, :V] = post.reshape(B * T, V)
    lens = torch.from_numpy(z["lens"]).to(I32)
    (fc, bc), (fg, bg) = run_pair(hip, fake, "psd_frame_stats", [p2, lens, torch.zeros(B * T, dtype=I32), torch.zeros(B * T), B, T, T, V, 0], [2, 3])
    assert torch.equal(fc, fg) and torch.equal(bc, bg)
    outs = run_pair(hip, fake, "psd_plan", [fc, bc, lens, torch.zeros(B * T, dtype=I32), torch.zeros(B * T, dtype=I32),
                                            torch.zeros(B, dtype=I32), B, T, 0, 0.9], [3, 4, 5])
    (sc, lc, nc), (sg, lg_, ng) = outs
    assert torch.equal(nc, ng) and np.array_equal(nc.numpy(), z["new_lens"])
    for b in range(B):
        n = int(nc[b])
        assert torch.equal(sc.view(B, T)[b, :n], sg.view(B, T)[b, :n]) and torch.equal(lc.view(B, T)[b, :n], lg_.view(B, T)[b, :n])
    Tout = int(nc.max())
    (oc,), (og,) = run_pair(hip, fake, "psd_gather", [p2, sc, lc, nc, torch.ones(B * Tout, ld), B, T, T, Tout, V], [4])
    assert rel_err(og, oc) < 1e-6
    ref = torch.from_numpy(z["out"])                            # the REAL reference's PSD output
    torch.testing.assert_close(og.view(B, Tout, ld)[:, :, :V], ref, rtol=1e-5, atol=1e-7)


def test_psd_kernels_random_batches_vs_reference(hip):
    """The three PSD kernels on the 32 random batches of tests/golden/psd_random.npz (outputs of the REAL reference's psd()):
    lengths exact, merged rows within fp32 rounding of the reference's running means."""
    from conftest import load_npz
    z = load_npz("psd_random")
    for c in range(int(z["n_cases"])):
        post = torch.from_numpy(z[f"c{c}_posterior"])
        B, T, V = post.shape
        ld = 64
        p2 = torch.zeros(B * T, ld)
        p2[:, :V] = post.reshape(B * T, V)
        p2, lens = p2.cuda(), torch.from_numpy(z[f"c{c}_lens"]).to(I32).cuda()
        fid, fbl = torch.zeros(B * T, dtype=I32, device="cuda"), torch.zeros(B * T, device="cuda")
        ss, sl = torch.zeros(B * T, dtype=I32, device="cuda"), torch.zeros(B * T, dtype=I32, device="cuda")
        nl = torch.zeros(B, dtype=I32, device="cuda")
        hip.psd_frame_stats(p2, lens, fid, fbl, B, T, T, V, 0)
        hip.psd_plan(fid, fbl, lens, ss, sl, nl, B, T, 0, 0.9)
        want_lens = z[f"c{c}_new_lens"]
        assert np.array_equal(nl.cpu().numpy(), want_lens), c
        Tout = int(want_lens.max())
        if Tout == 0:
            continue
        rows = torch.ones(B * Tout, ld, device="cuda")
        hip.psd_gather(p2, ss, sl, nl, rows, B, T, T, Tout, V)
        torch.testing.assert_close(rows.view(B, Tout, ld)[:, :, :V].cpu(), torch.from_numpy(z[f"c{c}_out"]), rtol=1e-5, atol=1e-7)


def test_psd_from_logits_equals_the_posterior_path(hip, fake):
    """Round 4: PSD works from the CTC head's bf16 logits (tasu_psd_logit_stats -> tasu_psd_plan -> tasu_psd_gather_softmax) and the
    fp32 posterior of all frames is never written.  Against the posterior path (softmax_rows -> psd_frame_stats -> psd_plan ->
    psd_gather) on peaky random logits with a strong blank: the same frame ids and lengths, blank probabilities and merged rows
    within fp32 rounding; ragged lengths, a ld wider than V, an all-blank utterance (zero rows); and against the CPU double."""
    g = torch.Generator().manual_seed(11)
    B, T, V, ld = 5, 97, 203, 256
    x = torch.randn(B, T, V, generator=g) * 2.0
    seg = torch.randint(0, V, (B, 20), generator=g).repeat_interleave(5, dim=1)[:, :T]
    x.scatter_add_(2, seg[..., None], torch.full((B, T, 1), 9.0))
    x[:, :, 0] += torch.where(torch.rand(B, T, generator=g) < 0.5, 12.0, 0.0)     # blank frames
    x[3, :, 0] += 40.0                                                             # an utterance PSD removes completely
    logits = torch.zeros(B * T, ld, dtype=BF)
    logits[:, :V] = x.reshape(B * T, V).to(BF)
    lens = torch.tensor([97, 60, 1, 97, 33], dtype=I32)
    lg, ln = logits.cuda(), lens.cuda()
    z32 = lambda *s_: torch.zeros(*s_, device="cuda")
    zi = lambda *s_: torch.zeros(*s_, dtype=I32, device="cuda")
    # posterior path
    post = z32(B * T, ld)
    hip.softmax_rows(lg, post, B * T, V)
    fid0, fbl0, ss0, sl0, nl0 = zi(B * T), z32(B * T), zi(B * T), zi(B * T), zi(B)
    hip.psd_frame_stats(post, ln, fid0, fbl0, B, T, T, V, 0)
    hip.psd_plan(fid0, fbl0, ln, ss0, sl0, nl0, B, T, 0, 0.9)
    # logits path
    fid1, fbl1, fst, ss1, sl1, nl1 = zi(B * T), z32(B * T), z32(B * T, 2), zi(B * T), zi(B * T), zi(B)
    hip.psd_logit_stats(lg, ln, fid1, fbl1, fst, B, T, T, V, 0)
    hip.psd_plan(fid1, fbl1, ln, ss1, sl1, nl1, B, T, 0, 0.9)
    torch.cuda.synchronize()
    assert torch.equal(fid0, fid1) and float((fbl0 - fbl1).abs().max()) < 1e-6
    assert torch.equal(nl0, nl1) and int(nl1[3]) == 0 and int(nl1.max()) > 10
    n = nl1.cpu()
    for b in range(B):
        assert torch.equal(ss0.view(B, T)[b, : n[b]], ss1.view(B, T)[b, : n[b]]) and torch.equal(sl0.view(B, T)[b, : n[b]], sl1.view(B, T)[b, : n[b]])
    Tout = int(n.max())
    r0, r1 = torch.ones(B * Tout, ld, device="cuda"), torch.ones(B * Tout, ld, device="cuda")
    hip.psd_gather(post, ss0, sl0, nl0, r0, B, T, T, Tout, V)
    hip.psd_gather_softmax(lg, fst, ss1, sl1, nl1, r1, B, T, T, Tout, V)
    torch.cuda.synchronize()
    assert float((r0 - r1).abs().max()) < 2e-6 and float(r1[:, V:].abs().max()) == 0.0
    assert float(r1.view(B, Tout, ld)[3].abs().max()) == 0.0 and abs(float(r1.view(B, Tout, ld)[0, 0].sum()) - 1.0) < 1e-4
    # the CPU double
    fidc, fblc, fstc, ssc, slc, nlc = torch.zeros(B * T, dtype=I32), torch.zeros(B * T), torch.zeros(B * T, 2), torch.zeros(B * T, dtype=I32), torch.zeros(B * T, dtype=I32), torch.zeros(B, dtype=I32)
    fake.psd_logit_stats(logits, lens, fidc, fblc, fstc, B, T, T, V, 0)
    fake.psd_plan(fidc, fblc, lens, ssc, slc, nlc, B, T, 0, 0.9)
    rc = torch.ones(B * Tout, ld)
    fake.psd_gather_softmax(logits, fstc, ssc, slc, nlc, rc, B, T, T, Tout, V)
    assert torch.equal(fidc, fid1.cpu()) and torch.equal(nlc, n) and float((rc - r1.cpu()).abs().max()) < 2e-6


# ------------------------------------------------------------------------------------------------ decode loop
def test_decode_kernels(hip, fake):
    B, S, H, G, nb, ctx = 2, 40, 4, 2, 3, 64
    M, LD, W = B * nb, (H + 2 * G) * HD, G * HD
    qkv_p = randn(B * S, LD, dtype=BF, seed=1)
    kc, vc = torch.zeros(M * ctx * W, dtype=BF), torch.zeros(M * ctx * W, dtype=BF)
    (kc1, vc1), (kg, vg) = run_pair(hip, fake, "kv_fill", [qkv_p, kc, vc, B, S, H, G, nb, ctx], [1, 2])
    assert torch.equal(kc1, kg) and torch.equal(vc1, vg)
    qkv = randn(M, LD, dtype=BF, seed=2)
    pos = torch.full((M,), S, dtype=I32)
    (kc2, vc2), (kg, vg) = run_pair(hip, fake, "kv_append", [qkv, kc1, vc1, pos, M, H, G, ctx], [1, 2])
    assert torch.equal(kc2, kg) and torch.equal(vc2, vg)
    kstart = torch.tensor([0, 0, 0, 5, 5, 5], dtype=I32)
    lens = torch.full((M,), S + 1, dtype=I32)
    (ix0,), (ixg,) = run_pair(hip, fake, "kv_index_init", [torch.zeros(M * ctx, dtype=I32), B, nb, S, ctx], [0])
    assert torch.equal(ix0, ixg) and int(ix0.view(M, ctx)[4, 0]) == 3 and int(ix0.view(M, ctx)[4, S]) == 4
    src = torch.tensor([1, 0, 0, 5, 3, 3], dtype=I32)
    (ix1,), (ixg,) = run_pair(hip, fake, "kv_index_reorder", [ix0, ix0.clone(), src, lens, M, ctx], [1])
    assert torch.equal(ix1, ixg) and int(ix1.view(M, ctx)[0, S]) == 1 and int(ix1.view(M, ctx)[0, S + 1]) == 0
    for ix in (ix0, ix1, None):
        (oc,), (og,) = run_pair(hip, fake, "attn_decode", [qkv, kc2, vc2, ix, kstart, lens, torch.zeros(M, H * HD, dtype=BF), M, H, G,
                                                           ctx, HD ** -0.5], [6])
        assert rel_err(og, oc) < 2e-2
    V, ld = 1000, 1024
    lg = torch.zeros(M, ld, dtype=BF)
    lg[:, :V] = randn(M, V, dtype=BF, seed=3, scale=3.0)
    banned = torch.tensor([int(lg[0, :V].float().argmax())], dtype=I32)
    for nban in (0, 1):
        (vc_, ic_), (vg_, ig_) = run_pair(hip, fake, "logprob_topk", [lg, M, V, 8, banned, nban, torch.zeros(M, 8), torch.zeros(M, 8, dtype=I32)],
                                          [6, 7])
        assert torch.equal(ic_, ig_), (ic_, ig_)
        assert float((vc_ - vg_).abs().max()) < 1e-4
    table = randn(50, 256, seed=4)
    ids = torch.randint(0, 50, (M,), dtype=I32)
    (c,), (g,) = run_pair(hip, fake, "embed_rows", [table, ids, torch.zeros(M, 256), M, 256], [2])
    assert torch.equal(c, g)


def test_logprob_topk_full_vocab(hip, fake):
    M, V = 4, 151936
    lg = randn(M, V, dtype=BF, seed=5, scale=3.0)
    (vc_, ic_), (vg_, ig_) = run_pair(hip, fake, "logprob_topk", [lg, M, V, 8, torch.zeros(1, dtype=I32), 0, torch.zeros(M, 8),
                                                                  torch.zeros(M, 8, dtype=I32)], [6, 7])
    assert torch.equal(ic_, ig_) and float((vc_ - vg_).abs().max()) < 1e-4


@pytest.mark.parametrize("M,N,K,mode,bias", [(64, 1536, 8960, 2, False), (64, 2048, 1536, 0, True), (40, 17920, 1536, 0, False),
                                             (64, 1000, 256, 1, True), (4, 151936, 1536, 0, False), (64, 1536, 1536, 2, False),
                                             (64, 5000, 3584, 0, True), (33, 3584, 3584, 2, False), (64, 3584, 18944, 2, False)])
def test_gemm_skinny(hip_both, fake, M, N, K, mode, bias):
    hip = hip_both
    ldc = (N + 63) // 64 * 64
    a = randn(M, K, dtype=BF, seed=1)
    b = randn(N, K, dtype=BF, seed=2, scale=1.0 / math.sqrt(K))
    bv = randn(N, dtype=BF, seed=3) if bias else None
    c = torch.zeros(M, ldc, dtype=BF if mode == 0 else F32)
    r = randn(M, ldc, seed=4) if mode == 2 else None
    cc, gc = c.clone(), c.cuda()
    ws = torch.zeros(32 * 64 * ((N + 95) // 96 * 96)).cuda()
    fake.gemm_skinny(a, b, cc, M, N, K, None, bias=bv, resid=r, mode=mode)
    ad, bd = a.cuda(), b.cuda()
    hip.gemm_skinny(ad, bd, gc, M, N, K, ws, bias=dev(bv), resid=dev(r), mode=mode)
    g2 = c.cuda()
    hip.gemm_skinny(ad, bd, g2, M, N, K, ws, bias=dev(bv), resid=dev(r), mode=mode)
    torch.cuda.synchronize()
    assert rel_err(gc, cc) < (1e-2 if mode != 1 else 2e-5 * math.sqrt(K))
    assert torch.equal(gc.cpu()[:, N:], cc[:, N:])
    assert torch.equal(gc, g2)


@pytest.mark.parametrize("M,I,K", [(64, 8960, 1536), (40, 200, 128), (64, 96, 4096), (64, 2400, 3584), (17, 18944, 3584)])
def test_gemm_skinny_swiglu(hip_both, fake, M, I, K):
    hip = hip_both
    a = randn(M, K, dtype=BF, seed=1)
    w = randn(2 * I, K, dtype=BF, seed=2, scale=1.0 / math.sqrt(K))
    ca, ga = torch.zeros(M, I, dtype=BF), torch.zeros(M, I, dtype=BF).cuda()
    ws = torch.zeros(32 * 64 * ((2 * I + 95) // 96 * 96)).cuda()
    fake.gemm_skinny_swiglu(a, w, ca, M, I, K, None)
    hip.gemm_skinny_swiglu(a.cuda(), w.cuda(), ga, M, I, K, ws)
    torch.cuda.synchronize()
    assert rel_err(ga, ca) < 2e-2


@pytest.mark.parametrize("M,N,K", [(64, 1536, 8960), (64, 1536, 1536), (33, 256, 512), (64, 17920, 128), (64, 3584, 18944), (40, 3584, 3584),
                                   (1, 3584, 18944)])
def test_gemm_skinny_norm(hip_both, fake, M, N, K):
    hip = hip_both
    a = randn(M, K, dtype=BF, seed=1)
    b = randn(N, K, dtype=BF, seed=2, scale=1.0 / math.sqrt(K))
    r = randn(M, N, seed=3)
    w = randn(N, seed=4).abs() + 0.5
    ws = torch.zeros(32 * 64 * ((N + 95) // 96 * 96)).cuda()
    (cc, yc), (cg, yg) = run_pair(hip, fake, "gemm_skinny_norm", [a, b, torch.zeros(M, N), r, M, N, K, w, torch.zeros(M, N, dtype=BF),
                                                                  1e-6, ws], [2, 8])
    assert rel_err(cg, cc) < 1e-2 and rel_err(yg, yc) < 2e-2


@pytest.mark.parametrize("M,N,K", [(64, 1536, 8960), (64, 1536, 1536), (33, 256, 512), (64, 256, 256), (1, 1536, 1536), (48, 1536, 8960),
                                   (7, 256, 1280)])
def test_gemm_stream_norm_in_one_launch_equals_the_two_launch_forms(hip, fake, M, N, K):
    """Round 5, tasu_gemm_stream_norm: the projection's workgroups store write-through, take a ticket, and the last arrivers
    normalise the finished rows inside the SAME launch (agent-scope hand-off, csrc/stream_body.h: norm_tail) -- against the
    two-launch forms (projection / slabs, then the norm kernel): the SAME BITS for the fp32 rows and the bf16 norm output, for
    one K range (the o projection) and K-range slabs (the down projection), ragged row counts, and 200 back-to-back launches plus
    a hipGraph replay loop (the ticket words must be back at zero after every launch)."""
    a = randn(M, K, dtype=BF, seed=1).cuda()
    b = randn(N, K, dtype=BF, seed=2, scale=1.0 / math.sqrt(K)).cuda()
    r = randn(M, N, seed=3).cuda()
    w = (randn(N, seed=4).abs() + 0.5).cuda()
    ws = torch.zeros(32 * 64 * ((N + 95) // 96 * 96)).cuda()

    def run(fused):
        hip.dec_fused_norm = fused
        c, y = torch.full((M, N), 7.0).cuda(), torch.full((M, N), 7.0, dtype=BF).cuda()
        hip.gemm_skinny_norm(a, b, c, r, M, N, K, w, y, 1e-6, ws)
        torch.cuda.synchronize()
        return c, y
    try:
        c0, y0 = run(False)
        c1, y1 = run(True)
        assert torch.equal(c0, c1) and torch.equal(y0, y1)
        assert int(hip.norm_sync.abs().sum()) == 0
        cc, yc = torch.zeros(M, N), torch.zeros(M, N, dtype=BF)
        fake.gemm_skinny_norm(a.cpu(), b.cpu(), cc, r.cpu(), M, N, K, w.cpu(), yc, 1e-6, None)
        assert rel_err(c1, cc) < 1e-2 and rel_err(y1, yc) < 2e-2
        hip.dec_fused_norm = True
        c, y = torch.zeros(M, N).cuda(), torch.zeros(M, N, dtype=BF).cuda()
        for i in range(200):                                                   # uneven arrival orders, launch after launch
            hip.gemm_skinny_norm(a, b, c, r, M, N, K, w, y, 1e-6, ws)
            if i % 50 == 49:
                torch.cuda.synchronize()
                assert torch.equal(c, c1) and torch.equal(y, y1) and int(hip.norm_sync.abs().sum()) == 0
        g = torch.cuda.CUDAGraph()
        y.zero_()
        with torch.cuda.graph(g):
            for _ in range(4):
                hip.gemm_skinny_norm(a, b, c, r, M, N, K, w, y, 1e-6, ws)
        for _ in range(25):
            g.replay()
        torch.cuda.synchronize()
        assert torch.equal(c, c1) and torch.equal(y, y1) and int(hip.norm_sync.abs().sum()) == 0
    finally:
        hip.dec_fused_norm = False


@pytest.mark.parametrize("M,H,G,K", [(64, 12, 2, 1536), (10, 4, 2, 256), (33, 2, 1, 128)])
def test_gemm_skinny_qkv_rope(hip_both, fake, M, H, G, K):
    hip = hip_both
    ctx, LD, W = 16, (H + 2 * G) * HD, G * HD
    a = randn(M, K, dtype=BF, seed=1)
    w = randn(LD, K, dtype=BF, seed=2, scale=1.0 / math.sqrt(K))
    bias = randn(LD, dtype=BF, seed=3)
    ang = randn(M, 64, seed=4)
    cos, sin = torch.cos(ang), torch.sin(ang)
    pos = (torch.arange(M) % ctx).to(I32)
    ws = torch.zeros(32 * 64 * ((LD + 95) // 96 * 96)).cuda()
    (qc, kc1, vc1), (qg, kg, vg) = run_pair(hip, fake, "gemm_skinny_qkv_rope",
                                            [a, w, bias, torch.zeros(M, LD, dtype=BF), M, H, G, K, cos, sin,
                                             torch.zeros(M * ctx * W, dtype=BF), torch.zeros(M * ctx * W, dtype=BF), pos, ctx, ws],
                                            [3, 10, 11])
    assert rel_err(qg, qc) < 2e-2 and rel_err(kg, kc1) < 2e-2 and rel_err(vg, vc1) < 2e-2
    assert torch.equal(kg != 0, kc1 != 0) or float(((kg != 0) != (kc1 != 0)).float().mean()) < 1e-3


def test_rope_append(hip, fake):
    M, H, G, ctx = 6, 4, 2, 16
    LD, W = (H + 2 * G) * HD, G * HD
    qkv = randn(M, LD, dtype=BF, seed=1)
    ang = randn(M, 64, seed=2)
    cos, sin = torch.cos(ang), torch.sin(ang)
    pos = torch.tensor([3, 0, 15, 7, 7, 1], dtype=I32)
    kc, vc = torch.zeros(M * ctx * W, dtype=BF), torch.zeros(M * ctx * W, dtype=BF)
    (qc, kc1, vc1), (qg, kg, vg) = run_pair(hip, fake, "rope_append", [qkv, cos, sin, kc, vc, pos, M, H, G, ctx], [0, 3, 4])
    assert rel_err(qg, qc) < 1e-2 and rel_err(kg, kc1) < 1e-2 and torch.equal(vg, vc1)
    assert torch.equal(kg != 0, kc1 != 0)


# ------------------------------------------------------------------------------------------------ device beam search
@pytest.mark.parametrize("quantise", [False, True])
@pytest.mark.parametrize("nb,lpw,min_len", [(4, 1.0, 1), (4, 2.0, 1), (3, 0.5, 4), (2, 1.0, 6), (1, 1.0, 1), (5, 1.0, 2)])
def test_beam_update_kernel_is_exact(hip, nb, lpw, min_len, quantise):
    """tasu_beam_update (selection, finished-hypothesis heap, early stop, back-pointers, next-step inputs) against the host
    BeamState on synthetic score streams -- 16 utterances, EOS events, exact ties when quantised: identical tokens, the same
    number of steps, and no state change from calls issued after it reported done."""
    from beam_stream import make_table, run_device, run_host
    B, V, T, eos = 16, 50, 14, 7
    table = make_table(3, V, eos, quantise)
    want = run_host(table, B, nb, T, eos, lpw, min_len)
    got, calls = run_device(hip, "cuda", table, B, nb, T, eos, lpw, min_len, extra_steps=3)
    fake, calls_f = run_device(FakeOps(), "cpu", table, B, nb, T, eos, lpw, min_len)
    assert np.array_equal(got, want), (got, want)
    assert np.array_equal(got, fake) and calls == calls_f


def unfrag(t, K):
    """[64, K] bf16 activation written in FRAGMENT ORDER ([K/32][4 row tiles][64 lanes][8]) back to row-major."""
    return t.reshape(-1)[:64 * K].view(K // 32, 4, 4, 16, 8).permute(1, 3, 0, 2, 4).reshape(64, K)


def ao_frag(a, K):
    """row-major [M <= 64, K] -> the fragment-order image of its 64-row chunk (rows >= M zero)."""
    full = torch.zeros(64, K, dtype=a.dtype)
    full[:a.shape[0]] = a
    return full.view(4, 16, K // 32, 4, 8).permute(2, 0, 3, 1, 4).contiguous().view(64, K)


@pytest.mark.parametrize("M,D,I,H,G,V", [(64, 1536, 8960, 12, 2, 4000), (40, 256, 512, 2, 1, 1000), (64, 512, 1792, 4, 2, 520),
                                         (1, 1536, 8960, 12, 2, 4000), (17, 1536, 8960, 12, 2, 700),
                                         (64, 3584, 18944, 28, 4, 2000), (23, 3584, 18944, 28, 4, 600)])
def test_decode_layer_chain_in_fragment_order(hip, fake, M, D, I, H, G, V):
    """One decode layer + lm_head on the streaming kernels with every operand in FRAGMENT ORDER (weights re-laid out by
    tasu_to_fragment_order, activations written in that order by the norm / attention / SwiGLU producers) against the same
    kernels on row-major operands: the layouts only change where bytes live, so every output must be BIT-identical; and both
    agree with the CPU double within bf16 tolerance."""
    ctx, LD, W, HHD = 24, (H + 2 * G) * HD, G * HD, H * HD
    g = torch.Generator().manual_seed(D + I)
    rn = lambda *sh, k=1.0: (torch.randn(*sh, generator=g) * k)
    wqkv, bq = rn(LD, D, k=D ** -0.5).to(BF), rn(LD).to(BF)
    wo, wgu, wd = rn(D, HHD, k=HHD ** -0.5).to(BF), rn(2 * I, D, k=D ** -0.5).to(BF), rn(D, I, k=I ** -0.5).to(BF)
    head = rn(V, D, k=D ** -0.5).to(BF)
    ln1, ln2, ln3 = 1 + 0.1 * rn(D), 1 + 0.1 * rn(D), 1 + 0.1 * rn(D)
    x0 = rn(M, D)
    ang = rn(M, 64)
    cos, sin = torch.cos(ang), torch.sin(ang)
    pos = (3 + torch.arange(M) % 5).to(I32)
    kc0, vc0 = rn(M * ctx * W).to(BF), rn(M * ctx * W).to(BF)
    kstart, lens = torch.zeros(M, dtype=I32), (pos + 1).to(I32)

    def chain(ops, dev_, frag):
        t = lambda a: a.to(dev_)
        w = dict(wqkv=t(wqkv), wo=t(wo), wgu=t(wgu), wd=t(wd), head=t(head))
        if frag:
            ops.register_decode_weight(w["wqkv"], "qkv", LD, H, G)
            ops.register_decode_weight(w["wo"], "plain", D)
            ops.register_decode_weight(w["wgu"], "swiglu", I)
            ops.register_decode_weight(w["wd"], "plain", D, slabs_ok=True)      # K = 8960: K-range slabs (18944: 12 x 1536 + 512)
            ops.register_decode_weight(w["head"], "plain", V)
            assert ops.begin_decode(D, HHD, I)
        Mp = 64
        x, x2 = t(x0.clone()), torch.zeros(M, D, device=dev_)
        xn, ao, act = (torch.zeros(Mp, n, dtype=BF, device=dev_) for n in (D, HHD, I))
        qkv = torch.zeros(M, LD, dtype=BF, device=dev_)
        kc, vc = t(kc0.clone()), t(vc0.clone())
        ws = torch.zeros(32 * 64 * ((max(V, 2 * I) + 95) // 96 * 96), device=dev_) if dev_ != "cpu" else None
        logits = torch.zeros(M, (V + 63) // 64 * 64, dtype=BF, device=dev_)
        # like ps_slm_amd/decode.py: row slices of buffers that hold a whole 64-row chunk (the base address is what counts)
        try:
            ops.dec_rmsnorm(x, t(ln1), xn[:M], 1e-6)
            ops.gemm_skinny_qkv_rope(xn[:M], w["wqkv"], t(bq), qkv, M, H, G, D, t(cos), t(sin), kc, vc, t(pos), ctx, ws)
            ops.attn_decode(qkv, kc, vc, None, t(kstart), t(lens), ao, M, H, G, ctx, HD ** -0.5)
            ops.gemm_skinny_norm(ao[:M], w["wo"], x2, x, M, D, HHD, t(ln2), xn[:M], 1e-6, ws)
            ops.gemm_skinny_swiglu(xn[:M], w["wgu"], act[:M], M, I, D, ws)
            ops.gemm_skinny_norm(act[:M], w["wd"], x, x2, M, D, I, t(ln3), xn[:M], 1e-6, ws)
            ops.gemm_skinny(xn[:M], w["head"], logits, M, V, D, ws)
            if dev_ != "cpu":
                torch.cuda.synchronize()
        finally:
            if frag:
                ops.end_decode()             # a failing call must not leave the next test in fragment order
        return [a.cpu() for a in (qkv, kc, vc, x2, x, logits[:, :V])]

    hip.use_stream = hip.dec_down_slabs = True             # both layouts sum the same K-range slabs (bit-identical results)
    row = chain(hip, "cuda", False)
    frg = chain(hip, "cuda", True)
    cpu = chain(fake, "cpu", False)
    for name, a, b, c in zip(("qkv", "kc", "vc", "x_mid", "x_out", "logits"), row, frg, cpu):
        assert torch.equal(a, b), name
        assert rel_err(a, c) < 3e-2, name


@pytest.mark.parametrize("M,D,HHD,I", [(64, 1536, 1536, 8960), (37, 1536, 1536, 8960), (64, 256, 256, 512), (1, 512, 512, 1792), (64, 1792, 1792, 2400)])
def test_post_attention_norm_inside_its_neighbours(hip, fake, M, D, HHD, I):
    """tasu_gemm_stream_resid_prenorm + tasu_gemm_stream_swiglu_rstd (o projection -> [norm] -> gate|up + SwiGLU with no norm launch)
    against the three-launch form (o projection + residual, RMSNorm, gate|up + SwiGLU) and the CPU double:
      * the fp32 residual stream c is BIT-identical (the projection's arithmetic is untouched);
      * the per-tile sums of squares add up to the rows' sums of squares of c (fp32 tolerance), rows >= M stay zero;
      * yw == bf16(norm_w * c) exactly;
      * the MLP activation agrees with the three-launch form like two bf16 evaluations of one expression (the rounding of the normed
        activation moves in front of the multiplication by rstd), in row-major and in fragment order (bit-identical to each other)."""
    g = torch.Generator().manual_seed(D + I + M)
    rn = lambda *sh, k=1.0: (torch.randn(*sh, generator=g) * k)
    ao, wo, wgu = rn(M, HHD).to(BF), rn(D, HHD, k=HHD ** -0.5).to(BF), rn(2 * I, D, k=D ** -0.5).to(BF)
    x, ln2 = rn(M, D) * 3.0, 1 + 0.1 * rn(D)
    outs = {}
    for frag in (False, True):
        w = dict(wo=wo.cuda(), wgu=wgu.cuda())
        hip.use_stream = True
        if frag:
            hip.register_decode_weight(w["wo"], "plain", D)
            hip.register_decode_weight(w["wgu"], "swiglu", I)
            wd = torch.zeros(D, I, dtype=BF, device="cuda")
            hip.register_decode_weight(wd, "plain", D, slabs_ok=True)
            assert hip.begin_decode(D, HHD, I)
        try:
            aod = torch.zeros(64, HHD, dtype=BF, device="cuda")
            if frag:
                aod.view(-1)[:] = ao_frag(ao, HHD).cuda().view(-1)
            else:
                aod[:M] = ao.cuda()
            c3, xn3, act3 = torch.zeros(M, D, device="cuda"), torch.zeros(64, D, dtype=BF, device="cuda"), torch.zeros(64, I, dtype=BF, device="cuda")
            hip.gemm_skinny_norm(aod[:M], w["wo"], c3, x.cuda(), M, D, HHD, ln2.cuda(), xn3[:M], 1e-6, None)
            hip.gemm_skinny_swiglu(xn3[:M], w["wgu"], act3[:M], M, I, D, None)
            c2, yw, act2 = torch.zeros(M, D, device="cuda"), torch.zeros(64, D, dtype=BF, device="cuda"), torch.zeros(64, I, dtype=BF, device="cuda")
            hip.dec_sumsq = None
            ssq = hip.gemm_skinny_prenorm(aod[:M], w["wo"], c2, x.cuda(), M, D, HHD, ln2.cuda(), yw[:M])
            hip.gemm_skinny_swiglu(yw[:M], w["wgu"], act2[:M], M, I, D, None, sumsq=ssq, eps=1e-6)
            torch.cuda.synchronize()
            act_frag = bool(hip.dec_frag_act)
        finally:
            if frag:
                hip.end_decode()
        assert torch.equal(c2, c3)
        part = ssq[:D // 16 * 64].view(D // 16, 64).cpu()
        want = (c2.cpu().double() ** 2).sum(1)
        assert torch.allclose(part[:, :M].double().sum(0), want, rtol=1e-5) and float(part[:, M:].abs().max() if M < 64 else 0.0) == 0.0
        ywr = unfrag(yw.cpu(), D)[:M] if frag else yw.cpu()[:M]
        assert torch.equal(ywr, (ln2 * c2.cpu()).to(BF))
        a2, a3 = (unfrag(t.cpu(), I)[:M] if act_frag else t.cpu()[:M] for t in (act2, act3))
        assert torch.isfinite(a2.float()).all() and rel_err(a2, a3) < 1.5e-2
        outs[frag] = (c2.cpu(), ywr, a2)
    for a, b in zip(outs[False], outs[True]):
        assert torch.equal(a, b)
    # the CPU double's three ops
    cc, xnc, actc = torch.zeros(M, D), torch.zeros(M, D, dtype=BF), torch.zeros(M, I, dtype=BF)
    fake.gemm_skinny_norm(ao, wo, cc, x, M, D, HHD, ln2, xnc, 1e-6, None)
    fake.gemm_skinny_swiglu(xnc, wgu, actc, M, I, D, None)
    assert rel_err(outs[True][2], actc) < 2e-2


@pytest.mark.parametrize("M,D,I,H,G", [(64, 1536, 8960, 12, 2), (23, 1536, 8960, 12, 2), (64, 512, 2560, 4, 2), (1, 1536, 8960, 12, 2)])
def test_input_norm_inside_the_slab_finish_and_qkv(hip, fake, M, D, I, H, G):
    """tasu_stream_finish_prenorm + tasu_gemm_stream_qkv_rope_rstd (down-projection slabs -> [input norm of the next layer] -> q|k|v + bias
    + RoPE + cache append, the norm without a whole-row kernel) against tasu_stream_finish_norm + tasu_gemm_stream_qkv_rope: the fp32
    residual stream is BIT-identical, yw == bf16(norm_w * c), the partial sums of squares add up to the rows', and q|k|v / the appended
    cache rows agree like two bf16 evaluations of one expression."""
    ctx, LD, W, HHD = 12, (H + 2 * G) * HD, G * HD, H * HD
    g = torch.Generator().manual_seed(D + I + M)
    rn = lambda *sh, k=1.0: (torch.randn(*sh, generator=g) * k)
    wqkv, bq, wd = rn(LD, D, k=D ** -0.5).to(BF).cuda(), rn(LD).to(BF).cuda(), rn(D, I, k=I ** -0.5).to(BF).cuda()
    wo, wgu = torch.zeros(D, HHD, dtype=BF, device="cuda"), torch.zeros(2 * I, D, dtype=BF, device="cuda")
    act, x2, ln = rn(M, I).to(BF), rn(M, D) * 3.0, 1 + 0.1 * rn(D)
    ang = rn(M, 64)
    cos, sin = torch.cos(ang).cuda(), torch.sin(ang).cuda()
    pos = (3 + torch.arange(M) % 5).to(I32).cuda()
    hip.use_stream = hip.dec_down_slabs = True
    hip.register_decode_weight(wqkv, "qkv", LD, H, G)
    hip.register_decode_weight(wo, "plain", D)
    hip.register_decode_weight(wgu, "swiglu", I)
    hip.register_decode_weight(wd, "plain", D, slabs_ok=True)
    assert hip.begin_decode(D, HHD, I)
    try:
        assert hip.prenorm_in_ok(D, I)
        actd = ao_frag(act, I).cuda()
        ws = torch.zeros(32 * 64 * ((2 * I + 95) // 96 * 96), device="cuda")
        outs = []
        for pre in (False, True):
            c, xn = torch.zeros(M, D, device="cuda"), torch.zeros(64, D, dtype=BF, device="cuda")
            qkv = torch.zeros(M, LD, dtype=BF, device="cuda")
            kc, vc = torch.zeros(M * ctx * W, dtype=BF, device="cuda"), torch.zeros(M * ctx * W, dtype=BF, device="cuda")
            ssq = hip.gemm_skinny_norm(actd[:M], wd, c, x2.cuda(), M, D, I, ln.cuda(), xn[:M], 1e-6, ws, prenorm_slot=0 if pre else None)
            assert (ssq is not None) == pre
            kw = dict(sumsq=ssq, eps=1e-6) if pre else {}
            hip.gemm_skinny_qkv_rope(xn[:M], wqkv, bq, qkv, M, H, G, D, cos, sin, kc, vc, pos, ctx, ws, **kw)
            torch.cuda.synchronize()
            outs.append((c.cpu(), unfrag(xn.cpu(), D)[:M], qkv.cpu(), kc.cpu(), vc.cpu(), None if ssq is None else ssq.cpu()))
    finally:
        hip.end_decode()
    (c3, xn3, q3, k3, v3, _), (c2, yw, q2, k2, v2, ssq) = outs
    assert torch.equal(c2, c3)
    assert torch.equal(yw, (ln * c2).to(BF))
    part = ssq[:D // 16 * 64].view(D // 16, 64)
    assert torch.allclose(part[:, :M].double().sum(0), (c2.double() ** 2).sum(1), rtol=1e-5)
    for a, b in ((q2, q3), (k2, k3), (v2, v3)):
        assert torch.isfinite(a.float()).all() and rel_err(a, b) < 1.5e-2
    assert ((k2 != 0) == (k3 != 0)).all()                     # the same cache slots were written


@pytest.mark.parametrize("H,G,ctx,frag", [(12, 2, 1100, 1), (2, 1, 700, 0), (28, 4, 530, 0), (12, 2, 40, 1)])
def test_attn_decode_long_ragged_contexts(hip, fake, H, G, ctx, frag):
    """Cache attention beyond what a wave prefetches (3 K chunks / 2 V blocks per wave = 384 / 512 keys): ragged visible ranges
    from 1 key to the whole context, random physical rows behind the index, 6 / 2 / 7 query heads per kv group, row-major and
    fragment-order outputs.  bf16 output: <= 2 % of the tensor's scale against the fp32 double."""
    M, W, LD = 9, G * HD, (H + 2 * G) * HD
    rs = np.random.RandomState(ctx)
    qkv = randn(M, LD, dtype=BF, seed=11)
    kc, vc = randn(M * ctx * W, dtype=BF, seed=12, scale=0.7), randn(M * ctx * W, dtype=BF, seed=13, scale=0.7)
    lens = np.array([1, 2, 17, ctx, ctx - 1, ctx // 2, 385, min(513, ctx), 33][:M], dtype=np.int32).clip(1, ctx)
    kstart = np.minimum(rs.randint(0, 9, size=M), lens - 1).astype(np.int32)
    index = torch.from_numpy(rs.randint(0, M, size=(M, ctx)).astype(np.int32))
    want = torch.zeros(M, H * HD, dtype=BF)
    fake.attn_decode(qkv, kc, vc, index, torch.from_numpy(kstart), torch.from_numpy(lens), want, M, H, G, ctx, HD ** -0.5)
    out = torch.zeros(64, H * HD, dtype=BF, device="cuda")
    hip.dec_frag = bool(frag)
    try:
        hip.attn_decode(dev(qkv), dev(kc), dev(vc), dev(index), dev(torch.from_numpy(kstart)), dev(torch.from_numpy(lens)), out, M, H, G,
                        ctx, HD ** -0.5)
    finally:
        hip.dec_frag = False
    torch.cuda.synchronize()
    got = out.cpu()
    if frag:
        K = H * HD
        got = got.view(K // 32, 4, 4, 16, 8).permute(1, 3, 0, 2, 4).reshape(64, K)
    assert torch.isfinite(got[:M].float()).all()
    assert rel_err(want, got[:M]) < 2e-2


@pytest.mark.parametrize("V,levels,k,nban", [(151936, 0, 8, 1), (151936, 5, 8, 2), (151936, 40, 16, 0), (5000, 2, 4, 3), (70, 0, 8, 1),
                                              (40000, 1, 6, 1)])
def test_logprob_topk_ties_and_full_vocabulary(hip, fake, V, levels, k, nban):
    """Token ids are index work: exact against the double, with the tie rule (smaller column first).  ``levels`` > 0 quantises
    the logits to that many distinct values, so that hundreds of columns tie at the selection threshold (the kernel's
    general path), 0 keeps random bf16 logits (its threshold path); banned ids include the row's best column."""
    M = 7
    ld = (V + 63) // 64 * 64
    g = torch.Generator().manual_seed(V + levels)
    x = torch.randn(M, V, generator=g) * 2.5
    if levels:
        x = torch.round(x.clamp(-3, 3) / 6 * levels) * (6 / max(levels, 1))
    lg = torch.zeros(M, ld, dtype=BF)
    lg[:, :V] = x.to(BF)
    banned = torch.tensor([int(lg[0, :V].float().argmax()), 3, -1][:max(nban, 1)], dtype=I32)
    (vc_, ic_), (vg_, ig_) = run_pair(hip, fake, "logprob_topk", [lg, M, V, k, banned, nban, torch.zeros(M, k), torch.zeros(M, k, dtype=I32)],
                                      [6, 7])
    assert torch.equal(ic_, ig_), (ic_, ig_)
    torch.testing.assert_close(vg_, vc_, rtol=0, atol=2e-4)


@pytest.mark.parametrize("D,M,nb,ctx", [(1536, 64, 4, 328), (256, 12, 3, 70), (1536, 100, 4, 200), (3584, 10, 5, 64)])
def test_decode_step_prologue_equals_its_five_launches(hip, D, M, nb, ctx):
    """Embedding rows + first input norm (fragment order) + RoPE factors + in-place beam reorder of the cache row index in one
    launch (tasu_decode_step_prologue) against tasu_embed_rows / tasu_rmsnorm_fwd_frag / tasu_rope_table / 2 x
    tasu_kv_index_reorder: bit for bit."""
    rs = np.random.RandomState(D + M)
    V = 500
    table = dev(randn(V, D, seed=1))
    ids = dev(torch.from_numpy(rs.randint(0, V, size=M).astype(np.int32)))
    w = dev(1 + randn(D, seed=2, scale=0.1))
    pos = dev(torch.from_numpy(rs.randint(0, 3000, size=M).astype(np.int32)))
    lens = dev(torch.from_numpy(rs.randint(1, ctx + 1, size=M).astype(np.int32)))
    groups = (np.arange(M) // nb) * nb
    src = dev(torch.from_numpy((groups + rs.randint(0, nb, size=M)).clip(max=M - 1).astype(np.int32)))     # a parent of the same utterance
    index0 = torch.from_numpy(rs.randint(0, M, size=(M, ctx)).astype(np.int32))
    Mp = (M + 63) // 64 * 64
    outs = []
    hip.dec_frag = True
    try:
        for fused in (False, True):
            hip.dec_prologue = fused
            x, xn = torch.zeros(M, D, device="cuda"), torch.zeros(Mp, D, dtype=BF, device="cuda")
            cos, sin = torch.zeros(M, 64, device="cuda"), torch.zeros(M, 64, device="cuda")
            index, tmp = dev(index0.clone()), dev(index0.clone())
            hip.decode_step_prologue(table, ids, x, w, xn, 1e-6, pos, cos, sin, HD, 1e6, index, tmp, src, lens, nb, M, D, ctx)
            torch.cuda.synchronize()
            outs.append((x, xn.view(torch.int16), cos, sin, index))
    finally:
        hip.dec_frag, hip.dec_prologue = False, True
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    want = index0.clone()
    for m in range(M):
        n = int(lens[m])
        want[m, :n] = index0[int(src[m]), :n]
    assert torch.equal(outs[1][4].cpu(), want)
