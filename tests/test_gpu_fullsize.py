"""Parity at BASELINE.json's FULL sizes (Qwen2.5-1.5B geometry of configs 2-4 and Qwen2.5-7B geometry of config 5, CTC
vocabulary 25,055, S = 256; the full 50 + 20-layer SenseVoiceSmall encoder at 504 frames), where the CPU oracle would take
minutes per step: size-independent properties of the path instead of a second implementation.

  * every decoder / lm_head / projector GEMM shape of the benchmark step against rocBLAS (torch.matmul) on the same bits;
  * the fused CE kernel at V = 151,936 against fp32 log-softmax of the very logits it consumed (loss, accuracy, dlogits);
  * run-to-run determinism of loss and gradients (bitwise);
  * the analytic projector gradient against a central finite difference of the loss along the gradient direction;
  * batch-permutation invariance and padding invariance of loss and gradients (ragged batch vs each utterance alone).

  * the full-size SANM encoder: a ragged batch equals every utterance alone, PSD lengths are exact.

Random-init weights of the real architecture (no checkpoint exists on the GPU box); inputs are seeded.
"""
import math
import zlib

import numpy as np
import pytest
import torch

from ps_slm_amd.model import Geometry, TasuModel
from ps_slm_amd.synthetic import synthetic_text_batch

pytestmark = pytest.mark.gpu
BF, F32 = torch.bfloat16, torch.float32


_MODELS = {}


def full_model(name):
    """One full-geometry model at a time on the device (the 7B one holds 2 x 15 GB of bf16 weights)."""
    from ps_slm_amd.ops import HipOps
    if name not in _MODELS:
        _MODELS.clear()
        torch.cuda.empty_cache()
        geo = Geometry.qwen25_1p5b() if name == "1.5b" else Geometry.qwen25_7b()
        m = TasuModel(geo, HipOps(), "cuda", keep_logits=True)
        m.init_random(seed=4321)
        _MODELS[name] = (geo, m)
    return _MODELS[name]


@pytest.fixture(scope="module")
def full():
    return full_model("1.5b")


def step(m, batch, backward=True):
    st = m.prepare_text(batch["input_ids"], batch["attention_mask"], batch["labels"], batch["post_ids"],
                        batch.get("alphas"), batch.get("keeps"))
    m.forward_projector_text(st)
    m.forward_llm(st, need_backward=backward)
    if backward:
        m.backward(st)
    torch.cuda.synchronize()
    return st


def take(batch, rows):
    """Sub-batch of utterances `rows`, trimmed to its own longest utterance (what the collator would have produced)."""
    am = batch["attention_mask"][rows]
    n = int(am.sum(1).max())
    out = dict(input_ids=batch["input_ids"][rows][:, :n], attention_mask=am[:, :n], labels=batch["labels"][rows][:, :n],
               post_ids=[batch["post_ids"][r] for r in rows])
    if "alphas" in batch:
        out["alphas"] = [batch["alphas"][r] for r in rows]
        out["keeps"] = [batch["keeps"][r] for r in rows]
    return out


# ------------------------------------------------------------------------------------------------ GEMM shapes
M_TOK = 4096
FULL_SHAPES = [("qkv", M_TOK, 2048, 1536), ("o", M_TOK, 1536, 1536), ("gate_up", M_TOK, 17920, 1536),
               ("down", M_TOK, 1536, 8960), ("d_down", M_TOK, 8960, 1536), ("d_gate_up", M_TOK, 1536, 17920),
               ("d_qkv", M_TOK, 1536, 2048), ("lm_head", M_TOK, 151936, 1536), ("d_lm_head", M_TOK, 1536, 151936),
               ("proj1", 1664, 2048, 25088), ("proj2", 1664, 1536, 2048), ("wgrad1", 2048, 25088, 1664),
               ("wgrad2", 1536, 2048, 1664), ("d_proj1", 1664, 25088, 2048)]


# Qwen2.5-7B (config 5): hidden 3584, 28 q / 4 kv heads, intermediate 18944, untied lm_head over 152064 ids
SHAPES_7B = [("7b_qkv", M_TOK, 4608, 3584), ("7b_o", M_TOK, 3584, 3584), ("7b_gate_up", M_TOK, 37888, 3584),
             ("7b_down", M_TOK, 3584, 18944), ("7b_d_down", M_TOK, 18944, 3584), ("7b_d_gate_up", M_TOK, 3584, 37888),
             ("7b_d_qkv", M_TOK, 3584, 4608), ("7b_lm_head", M_TOK, 152064, 3584), ("7b_d_lm_head", M_TOK, 3584, 152064),
             ("7b_proj2", 1664, 3584, 2048), ("7b_wgrad2", 3584, 2048, 1664), ("7b_d_proj2", 1664, 2048, 3584)]


@pytest.mark.parametrize("name,M,N,K", FULL_SHAPES + SHAPES_7B, ids=[s[0] for s in FULL_SHAPES + SHAPES_7B])
def test_benchmark_gemm_shapes_vs_rocblas(full, name, M, N, K):
    _, m = full
    g = torch.Generator(device="cuda").manual_seed(zlib.crc32(name.encode()) % 1000)
    a = torch.randn(M, K, generator=g, device="cuda").to(BF)
    b = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).to(BF)
    c = torch.empty(M, N, device="cuda", dtype=BF)
    m.ops.gemm(a, b, c, M, N, K)
    ref = torch.matmul(a, b.t())                       # rocBLAS bf16, fp32 accumulation
    torch.cuda.synchronize()
    diff = (c.float() - ref.float())
    scale = ref.float().abs().max()
    # both sides round an fp32 sum to bf16; the sums differ only by accumulation order -> at most one bf16 ulp apart
    assert float(diff.abs().max() / scale) < 2 ** -7
    assert float(diff.norm() / ref.float().norm()) < 2e-3
    c2 = torch.empty_like(c)
    m.ops.gemm(a, b, c2, M, N, K)
    torch.cuda.synchronize()
    assert torch.equal(c, c2)


@pytest.mark.parametrize("M,N,K,ksplit", [(2048, 1536, 151936, 2), (2048, 3584, 152064, 2), (1024, 1536, 151936, 2), (512, 1536, 65536, 4), (200, 256, 1024, 8)])
def test_splitk_gemm_vs_rocblas(full, M, N, K, ksplit):
    """The lm_head dgrad over the labelled rows (K = padded vocabulary) as ksplit K ranges per output tile + ordered fp32 sum +
    one bf16 rounding, against rocBLAS on the same bits; deterministic."""
    _, m = full
    g = torch.Generator(device="cuda").manual_seed(M + ksplit)
    a = (torch.randn(M, K, generator=g, device="cuda") * 0.05).to(BF)
    b = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).to(BF)
    ws = torch.empty(ksplit, M, N, device="cuda", dtype=F32)
    c = torch.empty(M, N, device="cuda", dtype=BF)
    m.ops.gemm_splitk(a, b, c, M, N, K, ksplit, ws)
    ref = torch.matmul(a, b.t())
    torch.cuda.synchronize()
    diff = c.float() - ref.float()
    assert float(diff.abs().max() / ref.float().abs().max()) < 2 ** -7
    assert float(diff.norm() / ref.float().norm()) < 2e-3
    c2 = torch.empty_like(c)
    m.ops.gemm_splitk(a, b, c2, M, N, K, ksplit, ws)
    torch.cuda.synchronize()
    assert torch.equal(c, c2)


# ------------------------------------------------------------------------------------------------ whole step
@pytest.mark.parametrize("size", ["1.5b", "7b"])
def test_full_size_step_properties(size):
    geo, m = full_model(size)
    batch = synthetic_text_batch(geo, 4, seed=99, noise=True, drop_prob=0.1, ragged=True)
    st = step(m, batch)
    loss = st.dev["loss_out"].clone()
    g = m.proj.g.clone()
    assert torch.isfinite(loss).all() and torch.isfinite(g).all()
    assert abs(float(loss[0]) - math.log(geo.llm_vocab)) < 1.0      # random init: CE near ln(V)

    # ---- fused CE at V = 151,936 against fp32 log-softmax of the same bf16 logits
    V = geo.llm_vocab
    logits = st.dev["logits"][:, :V].float()
    labels = st.dev["shift_labels"].long()
    sel = labels >= 0
    lp = torch.log_softmax(logits[sel], dim=-1)
    ref_loss = -lp.gather(1, labels[sel][:, None]).mean()
    assert abs(float(loss[0]) - float(ref_loss)) < 2e-5 * float(ref_loss) + 1e-5
    ref_hits = int((logits[sel].argmax(-1) == labels[sel]).sum())
    assert abs(float(loss[1]) * int(sel.sum()) - ref_hits) < 0.5
    assert int(loss[2]) == int(sel.sum())
    dl = st.dev["dlogits"][:, :V].float()
    ref_dl = torch.zeros_like(dl)
    p = lp.exp()
    p[torch.arange(p.shape[0], device="cuda"), labels[sel]] -= 1.0
    ref_dl[sel] = p / float(sel.sum())
    assert float((dl - ref_dl).norm() / ref_dl.norm()) < 4e-3        # bf16 storage of dlogits
    assert float(dl[~sel].abs().max()) == 0.0
    del logits, lp, p, dl, ref_dl

    # ---- determinism
    st2 = step(m, batch)
    assert torch.equal(st2.dev["loss_out"], loss) and torch.equal(m.proj.g, g)

    # ---- analytic gradient vs central finite difference along the gradient direction
    p0 = m.proj.p.clone()
    gn = float(g.norm())
    d = g / gn
    eps = 0.04 / gn                                    # predicted |dL| = 0.04 per side
    vals = []
    for sgn in (+1.0, -1.0):
        m.proj.p.copy_(p0 + sgn * eps * d)
        m.sync_projector_copies()
        vals.append(float(step(m, batch, backward=False).dev["loss_out"][0]))
    m.proj.p.copy_(p0)
    m.sync_projector_copies()
    fd = (vals[0] - vals[1]) / (2 * eps)
    assert abs(fd - gn) < 0.15 * gn, (fd, gn, vals)

    # ---- batch permutation: same loss and gradient up to summation order
    perm = [2, 0, 3, 1]
    stp = step(m, take(batch, perm))
    # Whole-tile GEMMs sum every row in the same order wherever the row sits (1.5B forward: the loss moves by the summation order
    # of the loss only).  The stream-K schedule (d_gate_up of both sizes at this batch's ~1000 rows; at 7B also down) cuts a tile's
    # K range where the tile's index puts it: a row's fp32 sums are then associated differently in another batch position, and
    # bf16 roundings flip through the 28 layers -- the same size of effect as the padding-invariance check below (7B loss ~ 1e-4
    # relative, gradients ~ 1 %)
    assert abs(float(stp.dev["loss_out"][0]) - float(loss[0])) < (2e-5 if size == "1.5b" else 5e-3)
    gp = m.proj.g
    assert float((gp - g).norm() / g.norm()) < 5e-2

    # ---- padding invariance: the ragged batch against every utterance alone (no padding at all)
    tot_loss, tot_cnt, gsum = 0.0, 0, torch.zeros_like(g)
    for r in range(4):
        s1 = step(m, take(batch, [r]))
        cnt = int(s1.dev["loss_out"][2])
        tot_loss += float(s1.dev["loss_out"][0]) * cnt
        tot_cnt += cnt
        gsum += m.proj.g * cnt
    assert tot_cnt == int(loss[2])
    assert abs(tot_loss / tot_cnt - float(loss[0])) < (1e-3 if size == "1.5b" else 3e-3)   # 7B: K = 18944 sums, other tile paths
    gsum /= tot_cnt
    cos = float(torch.nn.functional.cosine_similarity(gsum.flatten(), g.flatten(), dim=0))
    # two bf16 evaluations of the same gradient through 28 layers differ by ~3 % (rounding of dlogits / count differs)
    assert cos > 0.998 and float((gsum - g).norm() / g.norm()) < (5e-2 if size == "1.5b" else 8e-2)


def test_benchmark_shape_step_properties(full):
    """The EXACT step bench.py times (configs 2 / 3): 16 utterances x S = 256 (M = 4096 rows: the 96-tile stream-K plan of
    d_gate_up, the two-launch column split of gate|up, split-K lm_head dgrad), training mode -- loss head and the last layer's MLP
    on the 2,048 labelled rows only (keep_logits = False, tail_rows) -- plus the bucketed variable-shape batches of the
    `variable_S` sub-record.  Properties: bitwise run-to-run determinism; hipGraph replay == eager launches bitwise; the
    throughput schedule equals the all-rows full-logits schedule (loss / accuracy / count and gradients); CE of the all-rows run
    against fp32 log-softmax of its own logits; padding invariance of the bucketed shapes (the same batch unpadded)."""
    geo, m = full
    batch = synthetic_text_batch(geo, 16, seed=1234, noise=False)
    assert batch["input_ids"].shape[1] + 104 - 1 == 256

    def train_step(b, graphs=False, buckets=None):
        m.keep_logits, m.use_graphs, m.shape_buckets = False, graphs, buckets
        try:
            st = m.prepare_text(b["input_ids"], b["attention_mask"], b["labels"], b["post_ids"], b.get("alphas"), b.get("keeps"))
            m.run_forward_text(st)
            m.run_backward(st)
            torch.cuda.synchronize()
            return st, st.dev["loss_out"].clone(), m.proj.g.clone()
        finally:
            m.keep_logits, m.use_graphs, m.shape_buckets = True, False, None

    st, loss, g = train_step(batch)
    assert st.M == 4096 and st.S == 256 and st.nL == 2048 and "xout_tail" in st.dev      # the benchmark's shape and schedule
    assert torch.isfinite(loss).all() and torch.isfinite(g).all() and abs(float(loss[0]) - math.log(geo.llm_vocab)) < 1.0
    # the GEMM plans this shape takes (pinned on the CPU in tests/test_cabi.py; here: they ran)
    _, loss2, g2 = train_step(batch)
    assert torch.equal(loss, loss2) and torch.equal(g, g2)                               # determinism (stream-K sums in K order)
    for _ in range(3):                                                                   # eager warm-up, capture, replay
        _, loss3, g3 = train_step(batch, graphs=True)
    assert torch.equal(loss, loss3) and torch.equal(g, g3)                               # hipGraph replay == eager
    # all-rows schedule with materialised logits (what parity tests and eval use)
    sf = step(m, batch)
    lf, gf = sf.dev["loss_out"].clone(), m.proj.g.clone()
    assert abs(float(lf[0]) - float(loss[0])) < 2e-5 and int(lf[2]) == int(loss[2]) == 2048
    assert abs(float(lf[1]) - float(loss[1])) < 1e-6
    assert float((gf - g).norm() / gf.norm()) < 2e-2          # unlabelled rows' zero dlogits enter other stream-K / tile plans
    V = geo.llm_vocab
    labels = sf.dev["shift_labels"].long()
    sel = labels >= 0
    lp = torch.log_softmax(sf.dev["logits"][:, :V].float()[sel], dim=-1)
    ref_loss = -lp.gather(1, labels[sel][:, None]).mean()
    assert abs(float(lf[0]) - float(ref_loss)) < 2e-5 * float(ref_loss) + 1e-5
    del lp
    # ---- a ragged batch of the variable_S record, padded to the entrypoint's buckets vs unpadded
    rb = synthetic_text_batch(geo, 16, seed=77, noise=False, n_audio=88, target_len=101, ragged=True)
    _, l_pad, g_pad = train_step(rb, buckets=(16, 8, 256))
    _, l_raw, g_raw = train_step(rb)
    assert int(l_pad[2]) == int(l_raw[2]) and abs(float(l_pad[0]) - float(l_raw[0])) < 1e-3
    assert float((g_pad - g_raw).norm() / g_raw.norm()) < 5e-2


def test_full_geometry_audio_sft_step(full):
    """One config-4 step at full geometry (16 x 500 feature frames -> 70-layer SANM encoder -> CTC posterior -> device PSD ->
    projector -> LLM fwd / bwd): the PSD lengths of the three device kernels equal the oracle's PSD of the same posterior, the
    step is bitwise repeatable, and every utterance's posterior rows feed the LLM at the merge positions the plan says."""
    from oracle import tasu_oracle as O
    from ps_slm_amd.encoder import EncoderWeights
    geo, m = full
    if m.encoder is None:
        m.encoder = EncoderWeights(geo, m.device)
        m.encoder.init_random(4323)
    # a peaky posterior with a strong blank (the recipe of the encoder test below), so that PSD really merges runs and drops frames
    saved_w, saved_b = m.encoder.ctc_w.clone(), m.encoder.ctc_b.clone()
    m.encoder.ctc_w.mul_(6.0)
    m.encoder.ctc_b.zero_()
    m.encoder.ctc_b[geo.blank_id] = 4.0
    try:
        batch = synthetic_text_batch(geo, 16, seed=5, noise=False)
        lens = batch["input_feature_length"].clone()
        lens[3], lens[7] = 311, 97                                                   # ragged audio lengths
        m.keep_logits = False

        def run():
            st = m.prepare_audio(batch["input_ids"], batch["attention_mask"], batch["labels"], batch["input_features"], lens, do_psd=True)
            m.forward_llm(st)
            m.backward(st)
            torch.cuda.synchronize()
            return st
        st = run()
        loss, g = st.dev["loss_out"].clone(), m.proj.g.clone()
        assert torch.isfinite(loss).all() and torch.isfinite(g).all() and float(g.norm()) > 0
        T, Te, V = batch["input_features"].shape[1], batch["input_features"].shape[1] + 4, geo.ctc_vocab
        # the CTC head's bf16 logits prepare_audio left (the step never materialises the fp32 posterior): its softmax is the
        # posterior the reference's psd() would see
        logits = m._buf("enc_ctc_logits", (16 * Te, (V + 63) // 64 * 64), torch.bfloat16).view(16, Te, -1)
        post = torch.softmax(logits[:, :, :V].float(), -1)
        got = np.asarray(st.dev["psd_lens"])
        for b in (0, 3, 7, 15):
            body = post[b:b + 1, 4:, :V].cpu()
            _, want = O.psd(body, torch.tensor([int(lens[b])]), body)
            assert int(want[0]) == int(got[b]), (b, int(want[0]), int(got[b]))
        assert (got < np.asarray(lens)).all() and (got > 0).all()                  # merging / blank filtering really happened
        assert st.S == 24 + int(got.max()) + 128
        st2 = run()
        assert torch.equal(st2.dev["loss_out"], loss) and torch.equal(m.proj.g, g)
        # the frozen encoder one batch ahead (TasuModel.prefetch_encoder): the next pass runs on a side stream UNDER this step's
        # decoder GEMMs (256-CU grids on both streams, separate split-K workspaces) -- same bits, for the step it overlaps and
        # for the step that consumes it
        st3 = m.prepare_audio(batch["input_ids"], batch["attention_mask"], batch["labels"], batch["input_features"], lens, do_psd=True)
        m.forward_llm(st3)
        assert m.prefetch_encoder(batch["input_features"], lens)
        m.backward(st3)
        torch.cuda.synchronize()
        assert torch.equal(st3.dev["loss_out"], loss) and torch.equal(m.proj.g, g)
        assert m._enc_ahead is not None
        st4 = run()                                                                # consumes the prefetched pass
        assert m._enc_ahead is None
        assert torch.equal(st4.dev["loss_out"], loss) and torch.equal(m.proj.g, g) and np.array_equal(np.asarray(st4.dev["psd_lens"]), got)
    finally:
        m.encoder.ctc_w.copy_(saved_w)
        m.encoder.ctc_b.copy_(saved_b)
        m.keep_logits = True


# ------------------------------------------------------------------------------------------------ full-size encoder
def test_full_size_encoder_batch_equals_single_utterances():
    """SenseVoiceSmall at its published geometry (50 + 20 SANM layers, 4 x 128 heads, FSMN kernel 11, CTC vocabulary 25,055),
    500 + 4 frames, ragged lengths: every utterance's CTC posterior inside the padded batch must equal the posterior of that
    utterance run alone (key-padding mask, FSMN masking and row-independent GEMMs leave no cross-utterance path), rows are
    probability vectors, and the PSD lengths of the device kernels equal the oracle's PSD of the same posterior exactly."""
    from oracle import tasu_oracle as O
    from ps_slm_amd.encoder import encoder_posterior, psd_on_device
    from ps_slm_amd.ops import HipOps
    _MODELS.clear()
    torch.cuda.empty_cache()
    geo = Geometry.qwen25_1p5b()
    geo.llm_layers = 0                                   # encoder-only test: no decoder weights
    m = TasuModel(geo, HipOps(), "cuda")
    from ps_slm_amd.encoder import EncoderWeights
    from ps_slm_amd.synthetic import random_state_dict
    sd = {k: v for k, v in random_state_dict(geo, 77, with_encoder=True).items() if k.startswith("encoder.")}
    # a peaky head with a strong blank, so that PSD merges runs and drops blank frames (a flat random posterior keeps all)
    g = torch.Generator().manual_seed(3)
    sd["encoder.ctc.ctc_lo.weight"] = sd["encoder.ctc.ctc_lo.weight"] * 6.0
    bias = torch.zeros(geo.ctc_vocab)
    bias[geo.blank_id] = 4.0
    sd["encoder.ctc.ctc_lo.bias"] = bias
    m.encoder = EncoderWeights(geo, m.device)
    m.encoder.load_reference_state_dict(sd)
    T, lens = 500, [500, 377, 123]
    feats = torch.randn(3, T, geo.feat_dim, generator=g).half().float()
    for b, n in enumerate(lens):
        feats[b, n:] = 0
    V, Te = geo.ctc_vocab, T + 4
    post, _, _ = encoder_posterior(m, feats, torch.tensor(lens))
    torch.cuda.synchronize()
    post = post.view(3, Te, -1)[:, :, :V].clone()
    fl_dev = m._upload("feat_lens", np.asarray(lens, dtype=np.int32))
    _, new_lens, _ = psd_on_device(m, m._buf("enc_post", (3 * Te, (V + 63) // 64 * 64), F32), 3, T, Te, fl_dev, True)
    # rows are probability vectors
    for b, n in enumerate(lens):
        rows = post[b, : n + 4]
        assert torch.isfinite(rows).all() and float(rows.min()) >= 0.0
        assert float((rows.sum(-1) - 1).abs().max()) < 1e-4
    # PSD lengths: device kernels vs the oracle's PSD (restates ps-slm.py:237-317) on the same posterior
    body = post[:, 4:].cpu()
    _, want_lens = O.psd(body, torch.tensor(lens), body)
    assert np.array_equal(new_lens, want_lens.numpy()), (new_lens, want_lens)
    assert (new_lens < np.asarray(lens)).all() and (new_lens > 0).all()      # merging / blank filtering really happened
    # each utterance alone (no padding, own T) gives the same rows
    for b, n in enumerate(lens):
        alone, _, _ = encoder_posterior(m, feats[b:b + 1, :n], torch.tensor([n]))
        torch.cuda.synchronize()
        alone = alone.view(1, n + 4, -1)[0, :, :V]
        diff = float((alone - post[b, : n + 4]).abs().max())
        assert diff < 2e-3, (b, diff)                                        # probabilities; bf16 activations inside
        same_arg = (alone.argmax(-1) == post[b, : n + 4].argmax(-1)).float().mean()
        assert float(same_arg) > 0.995, (b, float(same_arg))


# ------------------------------------------------------------------------------------------------ decode
@pytest.mark.parametrize("size", ["1.5b", "7b"])
def test_full_size_decode_scores_equal_full_forward(size):
    """KV-cache decode == full recompute, at Qwen2.5-1.5B size (where the oracle would take an hour): the beam search's final
    score of every returned hypothesis (sum of the per-position log-probs its decode steps produced, / length) must equal the
    sum of log-softmax values a plain forward over [prompt + the returned tokens] assigns to those tokens.  That ties the cache
    fill / append, the beam reorder through the row index, the RoPE positions, the single-token attention, the weight-streaming
    GEMMs and the top-k log-probs to the training-forward kernels.  Tolerance 0.02 (1.5B) / 0.05 (7B) nats per token (two bf16
    evaluations of a 28-layer decoder; a wrong cache row or position moves a chosen token's log-prob by nats)."""
    from ps_slm_amd.decode import beam_search_generate
    geo, m = full_model(size)          # 7B: no streaming kernels for K = 3584 -> the split-K decode GEMMs, 7 query heads per kv group
    B, n_new, nb = 5, 7, 4
    batch = synthetic_text_batch(geo, B, seed=321, noise=False)
    ids = batch["input_ids"][:, :25]
    am = torch.ones_like(ids, dtype=torch.bool)
    st = m.prepare_text(ids, am, None, batch["post_ids"], None, None)
    m.forward_projector_text(st)
    out = beam_search_generate(m, st, num_beams=nb, max_new_tokens=n_new, eos_token_id=-1, pad_token_id=0)
    assert tuple(out.shape) == (B, n_new)
    beam_score = m._last_beam.fin_scores[:, 0].float().cpu() * n_new          # length_penalty 1.0: score = sum / length
    S = st.S
    ids2 = torch.cat([ids, out.to(ids.dtype)], dim=1)
    st2 = m.prepare_text(ids2, torch.ones_like(ids2, dtype=torch.bool), None, batch["post_ids"], None, None)
    m.forward_projector_text(st2)
    m.forward_llm(st2, compute_loss=False, need_backward=False)
    torch.cuda.synchronize()
    assert st2.S == S + n_new
    logits = m.logits_view(st2)
    lp = torch.log_softmax(logits[:, S - 1:S - 1 + n_new].float(), dim=-1)
    tok_lp = lp.gather(2, out.cuda().long()[:, :, None])[:, :, 0].cpu()
    assert float(tok_lp.min()) > -math.log(geo.llm_vocab) + 1.0             # the beams' tokens stand out of the uniform floor
    # measured: 1.5B 0.047 over the 7 tokens, 7B 0.22 (its chosen logits are 8-16, where a bf16 ulp is 0.06); another utterance's
    # tokens through the same forward: 25
    tol = (0.02 if size == "1.5b" else 0.05) * n_new
    assert float((tok_lp.sum(1) - beam_score).abs().max()) < tol, (tok_lp.sum(1), beam_score)
