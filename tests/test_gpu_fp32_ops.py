"""The fp32 decode family (csrc/fp32.hip, ``tasu_f32_*``) kernel by kernel against plain PyTorch fp32 / float64 on the same inputs.
Tolerances: these kernels round nowhere to bf16; what differs from torch is the order of fp32 sums, so results agree to a few
fp32 ulps of the largest term (1e-5 relative to the tensor's scale; top-k INDICES are exact).  The assembled path is pinned
token for token against the real reference in tests/test_gpu_model.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HD = 128


@pytest.fixture(scope="module")
def ops():
    from ps_slm_amd.ops import HipOps
    return HipOps()


def randn(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def close(a, b, tol=1e-5):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) < tol


@pytest.mark.parametrize("M,N,K", [(64, 1536, 1536), (64, 1536, 8960), (7, 2048, 1536), (64, 17920, 1536), (300, 520, 96), (1, 64, 32),
                                   (130, 200, 25088)])
@pytest.mark.parametrize("act,bias,resid", [(0, False, False), (1, True, False), (2, True, True), (0, True, True)])
def test_gemm_fp32(ops, M, N, K, act, bias, resid):
    """C = [resid +] act(A W^T + bias) against float64: tiles of 64 x 64 x 32 with ragged edges, one K range and K-range slabs
    (narrow outputs get a workspace and split), in-place residual."""
    a, w = randn(M, K, seed=1), randn(N, K, seed=2, scale=K ** -0.5)
    b = randn(N, seed=3) if bias else None
    r = randn(M, N, seed=4) if resid else None
    ref = a.double() @ w.double().t()
    if bias:
        ref = ref + b.double()
    if act == 1:
        ref = ref / (1 + torch.exp(-ref))
    elif act == 2:
        ref = ref.clamp_min(0)
    if resid:
        ref = ref + r.double()
    ws = torch.empty(16 * 128 * 4096, device="cuda")
    for use_ws in (False, True):
        c = r.clone() if resid else torch.full((M, N), float("nan"), device="cuda")
        ops.f32_gemm(a, w, c, M, N, K, bias=b, resid=c if resid else None, act=act, ws=ws if use_ws else None)   # resid aliases C
        torch.cuda.synchronize()
        assert close(c, ref, 2e-5), (use_ws, float((c.double() - ref).abs().max()))
    # the slab route is deterministic: two runs, the same bits
    c1, c2 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ops.f32_gemm(a, w, c1, M, N, K, bias=b, act=act, ws=ws)
    ops.f32_gemm(a, w, c2, M, N, K, bias=b, act=act, ws=ws)
    torch.cuda.synchronize()
    assert torch.equal(c1, c2)


@pytest.mark.parametrize("M,N,K,ks", [(64, 1536, 1536, 6), (64, 1536, 1536, 1), (33, 1000, 640, 5), (17, 48, 128, 1), (1, 16, 768, 6),
                                      (48, 2048, 256, 2), (64, 18992, 1536, 4), (16, 20, 384, 3), (64, 1536, 8960, 5), (64, 4000, 1536, 3),
                                      (32, 1536, 8960, 5), (64, 17920, 1536, 6)])
def test_gemm_fp32_at_beam_rows_streams_the_weights(ops, M, N, K, ks):
    """At most 64 rows: the weight-streaming kernel (f32_stream_kernel: register-resident activations staged through a wave-private
    LDS image, 16-column tiles walked by one workgroup per CU on a ring of three, waves split K, K ranges of 128 ks as slabs,
    hand-counted vmcnt).  Forced onto it with every ks (tasu_f32_gemm_stream; the dispatcher sends matrices >= 32 MB there: the
    last case and the lm_head).  Against float64: one K range (K = 128 ks) and slabs, ragged N (1000, 20: a partial last tile and
    a partial 4-column store), walks shorter and longer than the ring (1-75 tiles per workgroup), every row-block count, bias /
    activation / in-place residual; the same bits on every run; and a row alone has the bits it has among 64."""
    a, w = randn(M, K, seed=11), randn(N, K, seed=12, scale=K ** -0.5)
    b, r = randn(N, seed=13), randn(M, N, seed=14)
    ws = torch.empty(16 * 64 * N, device="cuda")
    base = a.double() @ w.double().t()
    for act, bias, resid in ((0, False, False), (1, True, True), (2, True, False)):
        ref = base + b.double() if bias else base
        ref = ref / (1 + torch.exp(-ref)) if act == 1 else (ref.clamp_min(0) if act == 2 else ref)
        ref = ref + r.double() if resid else ref
        c = r.clone() if resid else torch.full((M, N), float("nan"), device="cuda")
        ws.fill_(float("nan"))                                  # (nothing of an earlier call's slabs is read)
        ops.f32_gemm_stream(a, w, c, M, N, K, ks, bias=b if bias else None, resid=c if resid else None, act=act, ws=ws)
        torch.cuda.synchronize()
        assert close(c, ref, 2e-5), (act, float((c.double() - ref).abs().max()))
    c1, c2 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ops.f32_gemm_stream(a, w, c1, M, N, K, ks, ws=ws)
    ops.f32_gemm_stream(a, w, c2, M, N, K, ks, ws=ws)
    torch.cuda.synchronize()
    assert torch.equal(c1, c2)
    if M > 16:
        c3 = torch.empty(1, N, device="cuda")
        ops.f32_gemm_stream(a[M - 1:], w, c3, 1, N, K, ks, ws=ws)
        torch.cuda.synchronize()
        assert torch.equal(c3[0], c1[M - 1])
    # the dispatcher's choice (this kernel or the tile kernel) for the same problem: same product, and batch-independent too
    c4, c5 = torch.empty(M, N, device="cuda"), torch.empty(1, N, device="cuda")
    ops.f32_gemm(a, w, c4, M, N, K, ws=ws)
    ops.f32_gemm(a[M - 1:], w, c5, 1, N, K, ws=ws)
    torch.cuda.synchronize()
    assert close(c4, base, 2e-5) and torch.equal(c5[0], c4[M - 1])


@pytest.mark.parametrize("M,N,K,ks", [(64, 1536, 1536, 6), (33, 1000, 640, 5), (1, 16, 768, 2), (64, 17920, 1536, 6), (16, 1536, 8960, 5)])
def test_gemm_stream_on_fragment_order_weights(ops, M, N, K, ks):
    """tasu_f32_to_fragment_order lays a wave's operand pieces of every 16-row tile out contiguously ([tile][K / 16][64 lanes][4],
    rows past N zero); the streaming kernel on that copy gives the BITS it gives on the row-major matrix (same products, same order),
    through the forced entry and -- for a matrix of 32 MB and more -- through the dispatcher; a problem the streaming kernel does
    not serve refuses the copy, and HipOps.f32_weight hands such a problem the row-major matrix."""
    from ps_slm_amd.ops import TasuOpError
    a, w = randn(M, K, seed=21), randn(N, K, seed=22, scale=K ** -0.5)
    fr = ops.f32_to_fragments(w)
    torch.cuda.synchronize()
    T = (N + 15) // 16
    wp = torch.zeros(T * 16, K, device="cuda")
    wp[:N] = w
    want = wp.view(T, 16, K // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous().view(-1)      # [t][k16][g = lane >> 4][n = lane & 15][e]
    assert torch.equal(fr.t, want)
    ws = torch.empty(16 * 64 * N, device="cuda")
    c1, c2 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ops.f32_gemm_stream(a, w, c1, M, N, K, ks, ws=ws)
    ops.f32_gemm_stream(a, fr, c2, M, N, K, ks, ws=ws)
    torch.cuda.synchronize()
    assert torch.equal(c1, c2)
    served = ops.lib.tasu_f32_gemm_streams(M, N, K, ws.numel()) == 1
    assert served == (N * K * 4 >= 32 << 20)
    c3, c4 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ops.f32_gemm(a, w, c3, M, N, K, ws=ws)
    if served:
        ops.f32_gemm(a, fr, c4, M, N, K, ws=ws)
        assert ops.f32_weight(fr, M, ws) is fr
    else:
        with pytest.raises(TasuOpError):
            ops.f32_gemm(a, fr, c4, M, N, K, ws=ws)
        assert ops.f32_weight(fr, M, ws) is w
        ops.f32_gemm(a, ops.f32_weight(fr, M, ws), c4, M, N, K, ws=ws)
    torch.cuda.synchronize()
    assert torch.equal(c3, c4)


def test_gemm_stream_random_shapes(ops):
    """Forty random problems on the streaming kernel -- rows 1..64, ragged N, every ks with 1..16 K ranges, padded leading
    dimensions of both operands and of the output, row-major and fragment-order weights -- against float64, bits equal between
    the two weight layouts."""
    rng = np.random.default_rng(20261004)
    for case in range(40):
        M = int(rng.integers(1, 65))
        N = int(rng.integers(1, 2500))
        ks = int(rng.integers(1, 7))
        nsl = int(rng.integers(1, 17))
        K = 128 * ks * nsl
        lda, ldw, ldc = K + 4 * int(rng.integers(0, 3)), K + 4 * int(rng.integers(0, 3)), N + int(rng.integers(0, 5))
        a_full, w_full = randn(M, lda, seed=100 + case), randn(N, ldw, seed=200 + case, scale=K ** -0.5)
        a, w = a_full[:, :K], w_full[:, :K]
        c = torch.full((M, ldc), float("nan"), device="cuda")
        ws = torch.empty(16 * 64 * N, device="cuda")
        ops.f32_gemm_stream(a, w, c[:, :N], M, N, K, ks, ws=ws)
        fr = ops.f32_to_fragments(w)
        c2 = torch.full((M, ldc), float("nan"), device="cuda")
        ops.f32_gemm_stream(a, fr, c2[:, :N], M, N, K, ks, ws=ws)
        torch.cuda.synchronize()
        ref = a.double() @ w.double().t()
        assert close(c[:, :N], ref, 2e-5), (case, M, N, K, ks)
        assert torch.equal(c[:, :N], c2[:, :N]), (case, M, N, K, ks)
        assert torch.isnan(c[:, N:]).all()                      # nothing written past the N columns


def test_gemm_stream_refuses_what_it_cannot_serve(ops):
    from ps_slm_amd.ops import TasuOpError
    a, w, c = randn(65, 256), randn(32, 256), torch.empty(65, 32, device="cuda")
    ws = torch.empty(1 << 20, device="cuda")
    with pytest.raises(TasuOpError):
        ops.f32_gemm_stream(a, w, c, 65, 32, 256, 1, ws=ws)                 # 65 rows
    with pytest.raises(TasuOpError):
        ops.f32_gemm_stream(a[:8], w, c[:8], 8, 32, 256, 3, ws=ws)          # 256 is not a multiple of 384
    with pytest.raises(TasuOpError):
        ops.f32_gemm_stream(a[:8], w, c[:8], 8, 32, 256, 1, ws=None)        # two slabs, no workspace


@pytest.mark.parametrize("M,D", [(64, 1536), (5, 256), (130, 3584)])
def test_rmsnorm_fp32(ops, M, D):
    x, w = randn(M, D, seed=5, scale=3.0), randn(D, seed=6) + 1.0
    y = torch.empty_like(x)
    ops.f32_rmsnorm(x, w, y, M, D, 1e-6)
    ref = w.double() * (x.double() * torch.rsqrt(x.double().pow(2).mean(-1, keepdim=True) + 1e-6))
    assert close(y, ref)


@pytest.mark.parametrize("M,H,G", [(64, 12, 2), (9, 2, 1), (33, 28, 4)])
def test_rope_and_cache_append_fp32(ops, M, H, G):
    """q and k heads rotated in place like transformers' apply_rotary_pos_emb (x * cos + rotate_half(x) * sin, fp32), v untouched;
    with a cache the rotated k and the v rows land at cache[row, slot[row]]."""
    LD, W, ctx = (H + 2 * G) * HD, G * HD, 40
    qkv0 = randn(M, LD, seed=7)
    pos = torch.randint(0, 500, (M,), dtype=torch.int32).cuda()
    cos, sin = torch.empty(M, 64, device="cuda"), torch.empty(M, 64, device="cuda")
    ops.rope_table(pos, cos, sin, HD, 1e6)
    c, s = torch.cat([cos, cos], -1)[:, None, :], torch.cat([sin, sin], -1)[:, None, :]
    x = qkv0.view(M, H + 2 * G, HD).clone()
    rot = torch.cat([-x[..., 64:], x[..., :64]], -1)
    ref = x.clone()
    ref[:, : H + G] = x[:, : H + G] * c + rot[:, : H + G] * s
    qkv = qkv0.clone()
    ops.f32_rope(qkv, cos, sin, M, H, G)
    assert torch.equal(qkv.view(M, H + 2 * G, HD)[:, H + G:], x[:, H + G:])          # v: not touched
    assert torch.equal(qkv, ref.view(M, LD))                                          # the same roundings as torch eager: bit-identical
    kc, vc = torch.zeros(M * ctx * W, device="cuda"), torch.zeros(M * ctx * W, device="cuda")
    slot = torch.randint(0, ctx, (M,), dtype=torch.int32).cuda()
    qkv2 = qkv0.clone()
    ops.f32_rope(qkv2, cos, sin, M, H, G, kc, vc, slot, ctx)
    torch.cuda.synchronize()
    assert torch.equal(qkv2, qkv)
    kcv, vcv = kc.view(M, ctx, W), vc.view(M, ctx, W)
    rows = torch.arange(M).cuda()
    assert torch.equal(kcv[rows, slot.long()], qkv[:, H * HD:(H + G) * HD]) and torch.equal(vcv[rows, slot.long()], qkv[:, (H + G) * HD:])
    assert float(kc.abs().sum()) == pytest.approx(float(qkv[:, H * HD:(H + G) * HD].abs().sum()), rel=1e-5)     # nothing else written


def _attn_ref(q, k, v, allow, scale):
    """q [H, Sq, d], k / v [G, Sk, d], allow bool [Sq, Sk] -> [H, Sq, d] in float64 (GQA: head h uses KV head h // (H / G))."""
    H, G = q.shape[0], k.shape[0]
    rep = H // G
    kk, vv = k.double().repeat_interleave(rep, 0), v.double().repeat_interleave(rep, 0)
    sc = (q.double() @ kk.transpose(-1, -2)) * scale
    sc = sc.masked_fill(~allow[None], float("-inf"))
    return torch.softmax(sc, -1) @ vv


@pytest.mark.parametrize("B,S,H,G", [(2, 70, 12, 2), (1, 256, 2, 1), (3, 33, 28, 4), (2, 300, 4, 4), (5, 45, 4, 4), (2, 504, 4, 4)])
def test_attention_prefill_fp32(ops, B, S, H, G):
    """Causal prompt attention with left padding, and the bidirectional form with key lengths (the SANM encoder's)."""
    LD = (H + 2 * G) * HD
    qkv = randn(B * S, LD, seed=8)
    scale = HD ** -0.5
    kstart = torch.tensor([(7 * b) % max(S // 3, 1) for b in range(B)], dtype=torch.int32).cuda()
    out = torch.empty(B * S, H * HD, device="cuda")
    ops.f32_attn_prefill(qkv, kstart, out, B, S, H, G, scale)
    x = qkv.view(B, S, H + 2 * G, HD)
    for b in range(B):
        ks = int(kstart[b])
        q, k, v = x[b, :, :H].transpose(0, 1), x[b, :, H:H + G].transpose(0, 1), x[b, :, H + G:].transpose(0, 1)
        idx = torch.arange(S).cuda()
        allow = (idx[None, :] <= idx[:, None]) & (idx[None, :] >= ks)
        ref = _attn_ref(q, k, v, allow, scale).transpose(0, 1).reshape(S, H * HD)
        got = out.view(B, S, H * HD)[b]
        assert close(got[ks:], ref[ks:]), b
        assert float(got[:ks].abs().sum()) == 0.0                                         # padding rows: zeros
    klen = torch.tensor([S - (5 * b) % (S // 2) for b in range(B)], dtype=torch.int32).cuda()
    ops.f32_attn_prefill(qkv, None, out, B, S, H, G, scale, klen=klen)
    for b in range(B):
        n = int(klen[b])
        q, k, v = x[b, :, :H].transpose(0, 1), x[b, :, H:H + G].transpose(0, 1), x[b, :, H + G:].transpose(0, 1)
        allow = (torch.arange(S).cuda()[None, :] < n).expand(S, S)
        ref = _attn_ref(q, k, v, allow, scale).transpose(0, 1).reshape(S, H * HD)
        assert close(out.view(B, S, H * HD)[b, :n], ref[:n]), b
        if H == G:                                                                         # (the tiled kernel of the encoder's shape: zeros past the length)
            assert float(out.view(B, S, H * HD)[b, n:].abs().sum()) == 0.0


@pytest.mark.parametrize("B,nb,S,new,H,G", [(2, 4, 20, 9, 12, 2), (1, 3, 5, 30, 2, 1), (3, 2, 11, 4, 28, 4)])
def test_attention_decode_fp32_through_the_beam_index(ops, B, nb, S, new, H, G):
    """Single-token attention over the fp32 cache: key i of beam row m is read from cache row index[m, i] (the prompt lives in
    the utterance's first beam row, later positions wherever the reorders left them), keys [kstart, lens)."""
    M, ctx, W, LD = B * nb, S + new, G * HD, (H + 2 * G) * HD
    kc, vc = randn(M, ctx, W, seed=9), randn(M, ctx, W, seed=10)
    qkv = randn(M, LD, seed=11)
    g = torch.Generator().manual_seed(12)
    index = torch.empty(M, ctx, dtype=torch.int32)
    for m in range(M):
        b = m // nb
        index[m, :S] = b * nb
        index[m, S:] = torch.randint(b * nb, (b + 1) * nb, (new,), generator=g).int()
    index = index.cuda()
    kstart = torch.tensor([(3 * (m // nb)) % S for m in range(M)], dtype=torch.int32).cuda()
    lens = torch.tensor([S + 1 + (m % new) for m in range(M)], dtype=torch.int32).cuda()
    out = torch.empty(M, H * HD, device="cuda")
    scale = HD ** -0.5
    ops.f32_attn_decode(qkv, kc.view(-1), vc.view(-1), index, kstart, lens, out, M, H, G, ctx, scale)
    for m in range(M):
        lo, hi = int(kstart[m]), int(lens[m])
        pos = torch.arange(lo, hi).cuda()
        rows = index[m, lo:hi].long()
        k = kc[rows, pos].view(hi - lo, G, HD).transpose(0, 1)
        v = vc[rows, pos].view(hi - lo, G, HD).transpose(0, 1)
        q = qkv[m, : H * HD].view(H, 1, HD)
        ref = _attn_ref(q, k, v, torch.ones(1, hi - lo, dtype=torch.bool).cuda(), scale).reshape(H * HD)
        assert close(out[m], ref), m


def test_swiglu_embed_merge_kv_fill_fsmn_fp32(ops):
    M, I = 37, 520
    gu = randn(M, 2 * I, seed=13, scale=2.0)
    act = torch.empty(M, I, device="cuda")
    ops.f32_swiglu(gu, act, M, I)
    g, u = gu[:, :I].double(), gu[:, I:].double()
    assert close(act, g / (1 + torch.exp(-g)) * u, 1e-6)
    # embedding merge
    V, D, R = 50, 256, 9
    table, proj = randn(V, D, seed=14), randn(R, D + 64, seed=15)
    kind = torch.tensor([0, 1, 2, 2, 1, 0, 2], dtype=torch.int32).cuda()
    src = torch.tensor([0, 7, 3, 8, 49, 0, 0], dtype=torch.int32).cuda()
    x = torch.empty(7, D, device="cuda")
    ops.f32_embed_merge(table, proj, kind, src, x, 7, D)
    ref = torch.stack([torch.zeros(D).cuda() if k == 0 else (table[s] if k == 1 else proj[s, :D]) for k, s in zip(kind.tolist(), src.tolist())])
    assert torch.equal(x, ref)
    # prompt K / V into the cache rows of the first beams
    B, S, H, G, nb, ctx = 2, 6, 4, 2, 3, 10
    W, LD = G * HD, (H + 2 * G) * HD
    qkv = randn(B * S, LD, seed=16)
    kc, vc = torch.zeros(B * nb, ctx, W, device="cuda"), torch.zeros(B * nb, ctx, W, device="cuda")
    ops.f32_kv_fill(qkv, kc.view(-1), vc.view(-1), B, S, H, G, nb, ctx)
    for b in range(B):
        assert torch.equal(kc[b * nb, :S], qkv.view(B, S, LD)[b, :, H * HD:(H + G) * HD]) and torch.equal(vc[b * nb, :S], qkv.view(B, S, LD)[b, :, (H + G) * HD:])
    assert float(kc[1].abs().sum()) == 0.0
    # FSMN memory block (SenseVoice.py:124-140) against a conv1d restatement
    B, T, Dm, ks = 2, 23, 64, 11
    v = randn(B * T, 3 * Dm, seed=17)
    w = randn(Dm, ks, seed=18)
    lens = torch.tensor([23, 15], dtype=torch.int32).cuda()
    out0 = randn(B * T, Dm, seed=19)
    out = out0.clone()
    ops.f32_fsmn(v[:, 2 * Dm:], 3 * Dm, w, lens, out, B, T, Dm, ks)
    vv = v[:, 2 * Dm:].reshape(B, T, Dm).double()
    mask = (torch.arange(T).cuda()[None, :] < lens[:, None]).double()[..., None]
    xin = (vv * mask).transpose(1, 2)
    mem = torch.nn.functional.conv1d(torch.nn.functional.pad(xin, ((ks - 1) // 2, ks - 1 - (ks - 1) // 2)), w.double()[:, None, :], groups=Dm)
    ref = out0.double().view(B, T, Dm) + ((mem.transpose(1, 2) + vv * mask) * mask)
    assert close(out.view(B, T, Dm), ref)


@pytest.mark.parametrize("M,V,k", [(64, 151936, 8), (3, 1000, 2), (5, 70, 8), (2, 4097, 16), (4, 152064, 6), (7, 12, 16)])
def test_logprob_topk_fp32(ops, M, V, k):
    """(x - max) - log(sum exp(x - max)) of the k best selectable columns: values against torch.log_softmax in fp32, indices exact
    (ties: smaller column first), banned columns skipped; a row of massive ties takes the round-by-round form."""
    logits = randn(M, V, seed=20, scale=4.0)
    logits[0] = 0.5                                                                    # row 0: one winner, then V - 1 columns tied
    logits[0, 5] = 99.0                                                                # (more than the candidate list holds when V is large)
    banned = torch.tensor([int(logits[1 % M].argmax()), 3], dtype=torch.int32).cuda()
    ws = torch.empty(M * 16 * (2 + 2 * k), device="cuda")
    for nban, split in ((0, False), (2, False), (0, True), (2, True)):      # split: the row over 16 workgroups + a merge launch (workspace given)
        val, idx = torch.empty(M, k, device="cuda"), torch.empty(M, k, dtype=torch.int32, device="cuda")
        ws.fill_(float("nan"))
        ops.f32_logprob_topk(logits, M, V, k, banned, nban, val, idx, ws=ws if split else None)
        torch.cuda.synchronize()
        x64 = logits.double().cpu()
        lp32 = torch.log_softmax(logits, -1).cpu()
        if nban:
            x64[:, banned.cpu().long()] = float("-inf")
        # reference order: logit descending (log_softmax is monotone), column ascending
        order = torch.from_numpy(np.lexsort((np.arange(V)[None, :].repeat(M, 0), -x64.numpy()), axis=-1)[:, :k].copy())
        n_sel = V - nban
        kk = min(k, n_sel)
        assert torch.equal(idx.cpu().long()[:, :kk], order[:, :kk]), (nban, split)
        want = torch.gather(lp32, 1, order[:, :kk])
        assert float((val.cpu()[:, :kk] - want).abs().max()) < 2e-5


@pytest.mark.parametrize("M", [64, 5, 700])
def test_fused_slab_finishers_equal_the_two_kernel_forms(ops, M):
    """tasu_f32_gemm_resid_rmsnorm / _swiglu / _qkv_rope: the projection with the next row-wise kernel in the launch that sums its
    K-range slabs (M <= 64: the decode step) or, where the problem does not split (M = 700), as the two kernels -- the SAME BITS
    as tasu_f32_gemm_nt followed by tasu_f32_rmsnorm / tasu_f32_swiglu / tasu_f32_rope either way."""
    D, I, H, G, ctx = 1536, 2048, 12, 2, 16
    LD, W = (H + 2 * G) * HD, G * HD
    ws = torch.empty(16 * 128 * 4096, device="cuda")
    a = randn(M, H * HD, seed=30)
    wo, x0, nw = randn(D, H * HD, seed=31, scale=0.03), randn(M, D, seed=32), randn(D, seed=33) + 1.0
    # resid + norm
    x_ref = x0.clone()
    ops.f32_gemm(a, wo, x_ref, M, D, H * HD, resid=x_ref, ws=ws)
    y_ref = torch.empty(M, D, device="cuda")
    ops.f32_rmsnorm(x_ref, nw, y_ref, M, D, 1e-6)
    x, y = x0.clone(), torch.empty(M, D, device="cuda")
    ops.f32_gemm_resid_rmsnorm(a, wo, x, nw, y, M, D, H * HD, 1e-6, ws, resid=x)
    torch.cuda.synchronize()
    assert torch.equal(x, x_ref) and torch.equal(y, y_ref)
    # gate|up + SwiGLU
    h, wgu = randn(M, D, seed=34), randn(2 * I, D, seed=35, scale=0.03)
    gu, act_ref = torch.empty(M, 2 * I, device="cuda"), torch.empty(M, I, device="cuda")
    ops.f32_gemm(h, wgu, gu, M, 2 * I, D, ws=ws)
    ops.f32_swiglu(gu, act_ref, M, I)
    act = torch.empty(M, I, device="cuda")
    ops.f32_gemm_swiglu(h, wgu, torch.empty(M, 2 * I, device="cuda"), act, M, I, D, ws)
    torch.cuda.synchronize()
    assert torch.equal(act, act_ref)
    # q|k|v + bias + RoPE + cache append
    wqkv, bq = randn(LD, D, seed=36, scale=0.03), randn(LD, seed=37)
    pos = torch.randint(0, 300, (M,), dtype=torch.int32).cuda()
    cos, sin = torch.empty(M, 64, device="cuda"), torch.empty(M, 64, device="cuda")
    ops.rope_table(pos, cos, sin, HD, 1e6)
    slot = torch.randint(0, ctx, (M,), dtype=torch.int32).cuda()
    outs = []
    for fused in (False, True):
        qkv = torch.empty(M, LD, device="cuda")
        kc, vc = torch.zeros(M * ctx * W, device="cuda"), torch.zeros(M * ctx * W, device="cuda")
        if fused:
            ops.f32_gemm_qkv_rope(h, wqkv, bq, qkv, cos, sin, M, H, G, D, ws, kc=kc, vc=vc, slot=slot, ctx=ctx)
        else:
            ops.f32_gemm(h, wqkv, qkv, M, LD, D, bias=bq, ws=ws)
            ops.f32_rope(qkv, cos, sin, M, H, G, kc, vc, slot, ctx)
        outs.append((qkv, kc, vc))
    torch.cuda.synchronize()
    assert all(torch.equal(p, q) for p, q in zip(*outs))


# ------------------------------------------------------------------------------------------ fp32 training step: backward kernels
def test_rowwise_backward_kernels_fp32(ops):
    """RMSNorm / SwiGLU / SiLU backward, column sums, LayerNorm parameter gradients, transpose and row gather against torch autograd
    in float64 (csrc/fp32_train.hip)."""
    M, D, I = 37, 1536, 520
    x, w, dy = randn(M, D, seed=40, scale=2.0), randn(D, seed=41) + 1.0, randn(M, D, seed=42)
    xd = x.double().requires_grad_(True)
    y = w.double() * (xd * torch.rsqrt(xd.pow(2).mean(-1, keepdim=True) + 1e-6))
    (gx,) = torch.autograd.grad(y, xd, dy.double())
    dx0 = randn(M, D, seed=43)
    dx = dx0.clone()
    ops.f32_rmsnorm_bwd(dy, x, w, dx, M, D, 1e-6, True)
    assert close(dx, dx0.double() + gx)
    ops.f32_rmsnorm_bwd(dy, x, w, dx, M, D, 1e-6, False)
    assert close(dx, gx)
    # SwiGLU
    gu, dact = randn(M, 2 * I, seed=44, scale=2.0), randn(M, I, seed=45)
    gud = gu.double().requires_grad_(True)
    act = torch.nn.functional.silu(gud[:, :I]) * gud[:, I:]
    (ggu,) = torch.autograd.grad(act, gud, dact.double())
    dgu = torch.empty(M, 2 * I, device="cuda")
    ops.f32_swiglu_bwd(dact, gu, dgu, M, I)
    assert close(dgu, ggu)
    # SiLU forward / backward
    h = randn(M, I, seed=46, scale=3.0)
    out = torch.empty_like(h)
    ops.f32_silu(h, out)
    assert close(out, torch.nn.functional.silu(h.double()), 1e-6)
    hd = h.double().requires_grad_(True)
    (gh,) = torch.autograd.grad(torch.nn.functional.silu(hd), hd, dact.double())
    d_in = dact.clone()
    ops.f32_silu(h, d_in, dy=d_in)                                                      # in place
    assert close(d_in, gh)
    # column sums, transpose, gather
    cs = torch.empty(I, device="cuda")
    ops.f32_colsum(h, cs, M, I)
    assert close(cs, h.double().sum(0))
    Rp = 64
    tt = torch.full((I, Rp), float("nan"), device="cuda")
    ops.f32_transpose(h, tt, M, I, Rp)
    assert torch.equal(tt[:, :M], h.t()) and float(tt[:, M:].abs().sum()) == 0.0
    rows = torch.tensor([3, -1, 0, 36, -1, 7], dtype=torch.int32).cuda()
    g = torch.empty(6, D, device="cuda")
    ops.f32_gather_rows(x, rows, g, 6, D)
    assert torch.equal(g[0], x[3]) and torch.equal(g[3], x[36]) and float(g[1].abs().sum()) == 0.0 and float(g[4].abs().sum()) == 0.0
    # LayerNorm parameter gradients
    R, K, Kp = 50, 203, 256
    xin, dyn = randn(R, Kp, seed=47), randn(R, Kp, seed=48)
    gam, bet = randn(Kp, seed=49) + 1.0, randn(Kp, seed=50)
    yn = torch.empty(R, Kp, device="cuda")
    mean, rstd = torch.empty(R, device="cuda"), torch.empty(R, device="cuda")
    ops.layernorm_fwd(xin, gam, bet, yn, mean, rstd, R, K, 1e-5)
    gd, bd = gam[:K].double().requires_grad_(True), bet[:K].double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xin[:, :K].double(), (K,), gd, bd, 1e-5)
    assert close(yn[:, :K], yr)
    gg, gb = torch.autograd.grad(yr, (gd, bd), dyn[:, :K].double())
    dgam, dbet = torch.empty(K, device="cuda"), torch.empty(K, device="cuda")
    ops.f32_layernorm_bwd_params(dyn, xin, mean, rstd, dgam, dbet, R, K)
    assert close(dgam, gg) and close(dbet, gb)


@pytest.mark.parametrize("B,S,H,G", [(2, 70, 12, 2), (1, 130, 2, 1), (2, 33, 28, 4)])
def test_attention_backward_fp32(ops, B, S, H, G):
    """tasu_f32_attn_bwd (probabilities recomputed from the saved q|k|v, two launches) against autograd of the float64 attention,
    with left padding; and the rotation's backward = tasu_f32_rope(inverse)."""
    LD = (H + 2 * G) * HD
    qkv = randn(B * S, LD, seed=51)
    dout = randn(B * S, H * HD, seed=52)
    scale = HD ** -0.5
    kstart = torch.tensor([(5 * b) % max(S // 3, 1) for b in range(B)], dtype=torch.int32).cuda()
    dqkv = torch.full((B * S, LD), float("nan"), device="cuda")
    lse, delta = torch.empty(B * H * S, device="cuda"), torch.empty(B * H * S, device="cuda")
    ops.f32_attn_bwd(qkv, dout, kstart, dqkv, lse, delta, B, S, H, G, scale)
    for b in range(B):
        ks = int(kstart[b])
        x = qkv.view(B, S, H + 2 * G, HD)[b].double().requires_grad_(True)
        q, k, v = x[:, :H].transpose(0, 1), x[:, H:H + G].transpose(0, 1), x[:, H + G:].transpose(0, 1)
        idx = torch.arange(S).cuda()
        allow = (idx[None, :] <= idx[:, None]) & (idx[None, :] >= ks)
        o = _attn_ref(q[:, ks:], k, v, allow[ks:], scale).transpose(0, 1).reshape(S - ks, H * HD)   # (padding queries: no visible key, no gradient)
        do = dout.view(B, S, H * HD)[b].double()
        (gx,) = torch.autograd.grad(o, x, do[ks:])
        got = dqkv.view(B, S, H + 2 * G, HD)[b]
        assert close(got[ks:], gx[ks:], 2e-5), b
        assert float(got[:ks].abs().sum()) == 0.0
    # RoPE backward: the inverse rotation undoes the forward one (orthogonal), and equals autograd of the forward
    M = B * S
    pos = torch.randint(0, 400, (M,), dtype=torch.int32).cuda()
    cos, sin = torch.empty(M, 64, device="cuda"), torch.empty(M, 64, device="cuda")
    ops.rope_table(pos, cos, sin, HD, 1e6)
    y = qkv.clone()
    ops.f32_rope(y, cos, sin, M, H, G)
    ops.f32_rope(y, cos, sin, M, H, G, inverse=True)
    assert close(y, qkv, 1e-6)


def test_ce_gradient_fp32(ops):
    """tasu_f32_ce with dlogits (in place): the mean CE's gradient against autograd, pad columns zero, rows without a label zero."""
    M, V, ld = 9, 1000, 1024
    logits = randn(M, ld, seed=53, scale=3.0)
    labels = torch.tensor([5, -100, 999, 0, -100, 17, 3, 3, 500], dtype=torch.int32).cuda()
    n = int((labels >= 0).sum())
    inv = torch.tensor([1.0 / n], device="cuda")
    xd = logits[:, :V].double().requires_grad_(True)
    lab = labels.long().clone()
    loss = torch.nn.functional.cross_entropy(xd, lab, ignore_index=-100, reduction="mean")
    (gx,) = torch.autograd.grad(loss, xd)
    row_loss, row_hit = torch.empty(M, device="cuda"), torch.empty(M, dtype=torch.int32, device="cuda")
    buf = logits.clone()
    ops.f32_ce(buf, labels, M, V, row_loss, row_hit, dlogits=buf, inv_count=inv)
    torch.cuda.synchronize()
    assert close(buf[:, :V], gx, 1e-5) and float(buf[:, V:].abs().sum()) == 0.0
    assert abs(float(row_loss.sum()) / n - float(loss)) < 1e-5 and float(buf[1].abs().sum()) == 0.0
