"""End-to-end GPU parity of the training step at the kernel-compatible "mid" geometry:
HipOps (real kernels through the C-ABI) vs (a) the same host code on the FakeOps CPU double, (b) the oracle in
bf16-emulation mode, (c) golden vectors produced by the REAL reference in fp32 (bf16 tolerance stated)."""
import dataclasses

import numpy as np
import pytest
import torch

from conftest import load_npz
from fake_ops import FakeOps
from oracle import tasu_oracle as O
from ps_slm_amd.model import Geometry, TasuModel
from ps_slm_amd.synthetic import MID_GEOMETRY, random_state_dict, synthetic_text_batch

pytestmark = pytest.mark.gpu


def run_text(model, batch):
    st = model.prepare_text(batch["input_ids"], batch["attention_mask"], batch["labels"], batch["post_ids"],
                            batch.get("alphas"), batch.get("keeps"))
    model.forward_projector_text(st)
    model.forward_llm(st)
    model.backward(st)
    if model.device.type == "cuda":
        torch.cuda.synchronize()
    return st


@pytest.fixture(scope="module")
def setup():
    from ps_slm_amd.ops import HipOps
    geo = Geometry.from_dict(MID_GEOMETRY)
    sd = random_state_dict(geo, 2026, with_encoder=False)
    gm = TasuModel(geo, HipOps(), "cuda")
    gm.load_reference_state_dict(sd)
    cm = TasuModel(geo, FakeOps(), "cpu")
    cm.load_reference_state_dict(sd)
    return geo, sd, gm, cm


def cosine(a, b):
    return float(torch.nn.functional.cosine_similarity(a.flatten().float().cpu(), b.flatten().float().cpu(), dim=0))


@pytest.mark.parametrize("ragged,noise,drop", [(False, False, 0.0), (True, True, 0.15)])
def test_step_vs_double_and_oracle(setup, ragged, noise, drop):
    geo, sd, gm, cm = setup
    batch = synthetic_text_batch(geo, 3, seed=31, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=12,
                                 noise=noise, drop_prob=drop, ragged=ragged)
    sg, sc = run_text(gm, batch), run_text(cm, batch)
    lg, lc = sg.dev["loss_out"].cpu(), sc.dev["loss_out"]
    assert abs(float(lg[0]) - float(lc[0])) < 2e-3 and abs(float(lg[1]) - float(lc[1])) <= 1.0 / sc.plan.count + 1e-6
    valid = torch.from_numpy(sc.plan.key_mask[:, : sc.S].astype(bool))
    a, b = gm.logits_view(sg).float().cpu()[valid], cm.logits_view(sc).float()[valid]
    assert float((a - b).abs().max() / b.abs().max()) < 2e-2          # bf16 logits: a few ulps of the logit scale
    gg, gc = gm.projector_grads(), cm.projector_grads()
    for k in gc:
        assert cosine(gg[k], gc[k]) > 0.9995, k
        assert float((gg[k].cpu() - gc[k]).norm() / gc[k].norm()) < 3e-2, k
    out, grads = O.loss_and_projector_grads(sd, batch, dataclasses.asdict(geo), "bf16")
    assert abs(float(lg[0]) - float(out["loss"])) < 5e-3
    for k, g in grads.items():
        assert cosine(gg[k], g) > 0.999, k


def test_long_sequence_step_vs_double(setup):
    """S ~ 1000 (16 key tiles per query tile, ragged lengths): loss, logits and projector gradients against the CPU double."""
    geo, sd, gm, cm = setup
    batch = synthetic_text_batch(geo, 2, seed=41, prompt_len=30, n_audio=400, target_len=570, speech_pos=7, feat_frames=12,
                                 noise=True, drop_prob=0.05, ragged=True)
    sg, sc = run_text(gm, batch), run_text(cm, batch)
    assert sg.S == sc.S and sg.S > 900
    lg, lc = sg.dev["loss_out"].cpu(), sc.dev["loss_out"]
    assert abs(float(lg[0]) - float(lc[0])) < 2e-3 and abs(float(lg[1]) - float(lc[1])) <= 1.0 / sc.plan.count + 1e-6
    gg, gc = gm.projector_grads(), cm.projector_grads()
    for k in gc:
        assert cosine(gg[k], gc[k]) > 0.999, k


def test_step_vs_reference_golden(setup):
    """Golden = the real reference in fp32 (tests/golden/mid_text_clean.npz).  Stated bf16 tolerances:
    |loss - ref| <= 2e-2, logits max-abs error <= 3% of the logit range, projector grads cosine >= 0.995."""
    geo, sd, gm, _ = setup
    z = load_npz("mid_text_clean")
    assert int(z["seed_w"]) == 2026
    batch = synthetic_text_batch(geo, 3, seed=int(z["seed_b"]), prompt_len=9, n_audio=21, target_len=17, speech_pos=4,
                                 feat_frames=12, noise=True, drop_prob=0.15, ragged=True)
    batch["post_ids"] = [list(np.asarray(p)[np.asarray(k, dtype=bool)]) for p, k in zip(batch["post_ids"], batch["keeps"])]
    del batch["alphas"], batch["keeps"]
    st = run_text(gm, batch)
    res = st.dev["loss_out"].cpu()
    assert abs(float(res[0]) - float(z["loss"])) < 2e-2
    valid = torch.from_numpy(st.plan.key_mask[:, : st.S].astype(bool))
    cols = torch.from_numpy(z["cols"])
    lg = gm.logits_view(st).float().cpu()
    ref = torch.from_numpy(z["logits_cols"])
    assert float((lg[:, :, cols] - ref)[valid].abs().max() / ref[valid].abs().max()) < 3e-2
    gg = gm.projector_grads()
    for k, g in gg.items():
        short = "grad." + k[len("encoder_projector."):]
        if short in z:
            assert cosine(g, torch.from_numpy(z[short])) > 0.995, k


def test_adamw_step_moves_loss_down(setup):
    """A few optimizer steps on one batch must reduce the loss and keep the K-padding columns exactly zero."""
    geo, sd, gm, _ = setup
    batch = synthetic_text_batch(geo, 2, seed=9, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=12)
    lr = torch.tensor([1e-3], device="cuda")
    losses = []
    for step in range(1, 6):
        st = run_text(gm, batch)
        losses.append(float(st.dev["loss_out"][0]))
        gm.ops.adamw(gm.proj.p, gm.proj.g, gm.proj.m, gm.proj.v, gm.proj.pb, lr, 0.9, 0.999, 1e-6, 0.0, step, 1.0)
        gm.proj.refresh_working_copies(gm.ops)
    assert losses[-1] < losses[0]
    w1 = gm.proj.view(gm.proj.p, "ffn.0.weight")
    assert float(w1[:, geo.ctc_vocab:].abs().max()) == 0.0
    gm.load_reference_state_dict(sd)  # restore for other tests


# ------------------------------------------------------------------ audio path on the GPU
def run_audio(model, batch):
    st = model.prepare_audio(batch["input_ids"], batch["attention_mask"], batch["labels"], batch["input_features"],
                             batch["input_feature_length"])
    model.forward_llm(st)
    model.backward(st)
    if model.device.type == "cuda":
        torch.cuda.synchronize()
    return st


def test_audio_step_with_psd_vs_double_and_reference_golden():
    """SANM encoder -> CTC softmax -> PSD (merging + blank filter exercised) -> projector -> LLM, fwd + bwd.
    Golden = real reference fp32; PSD lengths must be bit-exact, loss within 3e-2, grads cosine >= 0.99."""
    from conftest import mid_audio_psd_case
    from ps_slm_amd.ops import HipOps
    geo, sd, batch, z = mid_audio_psd_case()
    gm = TasuModel(geo, HipOps(), "cuda")
    gm.load_reference_state_dict(sd)
    cm = TasuModel(geo, FakeOps(), "cpu")
    cm.load_reference_state_dict(sd)
    sg, sc = run_audio(gm, batch), run_audio(cm, batch)
    assert np.array_equal(sg.dev["psd_lens"], z["psd_lens"]) and np.array_equal(sc.dev["psd_lens"], z["psd_lens"])
    lg, lc = float(sg.dev["loss_out"][0]), float(sc.dev["loss_out"][0])
    assert abs(lg - lc) < 3e-3 and abs(lg - float(z["loss"])) < 3e-2
    a, b = sg.dev["post"][: sg.Ra].cpu(), sc.dev["post"][: sc.Ra]
    # PSD'd posterior rows are probabilities of bf16 CTC logits: one bf16 ulp of a logit near 8 (0.03) moves a
    # probability by at most 0.03 * p(1-p) <= 7.5e-3
    assert float((a - b).abs().max()) < 2e-2
    gg, gc = gm.projector_grads(), cm.projector_grads()
    for k in gc:
        assert cosine(gg[k], gc[k]) > 0.999, k
        short = "grad." + k[len("encoder_projector."):]
        if short in z:
            assert cosine(gg[k], torch.from_numpy(z[short])) > 0.99, k


def test_audio_step_without_merging(setup):
    geo = Geometry.from_dict(MID_GEOMETRY)
    from ps_slm_amd.ops import HipOps
    z = load_npz("mid_audio")
    sd = random_state_dict(geo, int(z["seed_w"]), with_encoder=True)
    batch = synthetic_text_batch(geo, 3, seed=int(z["seed_b"]), prompt_len=9, n_audio=21, target_len=17, speech_pos=4,
                                 feat_frames=12, noise=True, drop_prob=0.15, ragged=True)
    gm = TasuModel(geo, HipOps(), "cuda")
    gm.load_reference_state_dict(sd)
    st = run_audio(gm, batch)
    assert abs(float(st.dev["loss_out"][0]) - float(z["loss"])) < 3e-2
    gg = gm.projector_grads()
    for k, g in gg.items():
        short = "grad." + k[len("encoder_projector."):]
        if short in z:
            assert cosine(g, torch.from_numpy(z[short])) > 0.99, k


def test_generate_beam4_on_gpu():
    """Decode loop on the GPU (KV cache, cache attention, top-k kernel) vs the same host code on the CPU double and
    the reference's generate() tokens (fp32; bf16 may flip a near-tie late, so a common prefix >= 8 is required)."""
    from conftest import mid_audio_psd_case, split_flat
    from ps_slm_amd.decode import beam_search_generate
    from ps_slm_amd.ops import HipOps
    geo, sd, batch, _ = mid_audio_psd_case()
    z = load_npz("mid_generate_beam4")
    word_ids = split_flat(z["post_ids_flat"], z["post_lens"])
    ids, am = torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"])
    gm = TasuModel(geo, HipOps(), "cuda")
    gm.load_reference_state_dict(sd)
    cm = TasuModel(geo, FakeOps(), "cpu")
    cm.load_reference_state_dict(sd)
    outs = {}
    for name, m in (("gpu", gm), ("cpu", cm)):
        st = m.prepare_text(ids, am, None, word_ids, None, None)
        m.forward_projector_text(st)
        outs[name + "_text"] = beam_search_generate(m, st, max_new_tokens=16).numpy()
        st = m.prepare_audio(ids, am, None, batch["input_features"][:2], batch["input_feature_length"][:2])
        outs[name + "_audio"] = beam_search_generate(m, st, max_new_tokens=16).numpy()
    for path, key in (("text", "tokens_text"), ("audio", "tokens_audio")):
        g, c, r = outs["gpu_" + path], outs["cpu_" + path], z[key]
        assert ((g == c).cumprod(1).sum(1) >= 8).all(), (path, g, c)
        assert ((g == r).cumprod(1).sum(1) >= 8).all(), (path, g, r)


@pytest.mark.parametrize("ragged", [False, True])
def test_labelled_rows_loss_head_on_gpu(setup, ragged):
    """The training step's throughput mode (keep_logits=False: lm_head / CE / lm_head dgrad over the labelled rows only, CE
    writing dlogits over the logits) against full materialisation on the GPU: loss and accuracy equal to fp32 rounding of the
    row sums, count exact, projector gradients equal up to the accumulation order of the two GEMM row layouts."""
    from ps_slm_amd.ops import HipOps
    geo, sd, gm, _ = setup
    lean = TasuModel(geo, HipOps(), "cuda", keep_logits=False)
    lean.load_reference_state_dict(sd)
    batch = synthetic_text_batch(geo, 3, seed=9, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=12,
                                 noise=True, drop_prob=0.1, ragged=ragged)
    sf, sl = run_text(gm, batch), run_text(lean, batch)
    lf, ll = sf.dev["loss_out"].cpu(), sl.dev["loss_out"].cpu()
    assert float(lf[2]) == float(ll[2]) and float(lf[1]) == float(ll[1])
    assert abs(float(lf[0]) - float(ll[0])) < 1e-5 * float(lf[0])
    assert float((gm.proj.g - lean.proj.g).norm() / gm.proj.g.norm()) < 5e-3
    assert cosine(gm.proj.g, lean.proj.g) > 0.9999


@pytest.mark.parametrize("ragged", [False, True])
def test_tail_layer_on_labelled_rows_on_gpu(setup, ragged):
    """TasuModel.tail_rows on the HIP kernels against the all-rows schedule: same loss, accuracy and count, and projector
    gradients equal bit for bit (every output element of the compact GEMMs sees its K range in the same order; the row gathers /
    scatters move values only)."""
    from ps_slm_amd.ops import HipOps
    geo, sd, _, _ = setup
    a, b = TasuModel(geo, HipOps(), "cuda", keep_logits=False), TasuModel(geo, HipOps(), "cuda", keep_logits=False)
    a.load_reference_state_dict(sd)
    b.load_reference_state_dict(sd)
    a.tail_rows, b.tail_rows = True, False
    batch = synthetic_text_batch(geo, 3, seed=11, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=12,
                                 noise=True, drop_prob=0.1, ragged=ragged)
    sa, sb = run_text(a, batch), run_text(b, batch)
    assert "xout_tail" in sa.dev and "xout_tail" not in sb.dev
    assert torch.equal(sa.dev["loss_out"], sb.dev["loss_out"])
    assert torch.equal(a.proj.g, b.proj.g)


@pytest.mark.parametrize("k", [1, 2])
def test_linear_projector_on_gpu(k):
    """``encoder_projector="linear"`` (EncoderProjectorConcat, k frames per projector row) on the HIP kernels: against the REAL
    reference's fp32 golden (|loss - ref| <= 2e-2, logits <= 3 % of their range, gradient cosine >= 0.995) and against the
    same host code on the CPU double (loss 2e-3, gradient cosine 0.9995)."""
    from conftest import linear_projector_case
    from ps_slm_amd.ops import HipOps
    geo, sd, batch, z = linear_projector_case(k)
    gm = TasuModel(geo, HipOps(), "cuda")
    gm.load_reference_state_dict(sd)
    cm = TasuModel(geo, FakeOps(), "cpu")
    cm.load_reference_state_dict(sd)
    sg, sc = run_text(gm, batch), run_text(cm, batch)
    lg = float(sg.dev["loss_out"][0])
    assert abs(lg - float(z["loss"])) < 2e-2 and abs(lg - float(sc.dev["loss_out"][0])) < 2e-3
    valid = torch.from_numpy(sg.plan.key_mask[:, : sg.S].astype(bool))
    cols = torch.from_numpy(z["cols"])
    ref = torch.from_numpy(z["logits_cols"])
    assert float((gm.logits_view(sg).float().cpu()[:, :, cols] - ref)[valid].abs().max() / ref[valid].abs().max()) < 3e-2
    gg, gc = gm.projector_grads(), cm.projector_grads()
    for name in gc:
        assert cosine(gg[name], gc[name]) > 0.9995, name
    assert cosine(gg["encoder_projector.linear1.bias"], torch.from_numpy(z["grad.linear1.bias"])) > 0.995
    assert cosine(gg["encoder_projector.linear2.bias"], torch.from_numpy(z["grad.linear2.bias"])) > 0.995
    assert cosine(gg["encoder_projector.linear1.weight"][::64], torch.from_numpy(z["grad.linear1.weight.rows64"])) > 0.995


@pytest.mark.parametrize("k", [1, 2])
def test_cov1d_projector_on_gpu(k):
    """``encoder_projector="cov1d-linear"`` (EncoderProjectorCov1d: the convolution runs as one GEMM over k concatenated frames)
    on the HIP kernels: against the REAL reference's fp32 golden (|loss - ref| <= 2e-2, logits <= 3 % of their range, gradient
    cosine >= 0.995) and against the same host code on the CPU double (loss 2e-3, gradient cosine 0.9995)."""
    from conftest import cov1d_projector_case
    from ps_slm_amd.ops import HipOps
    geo, sd, batch, z = cov1d_projector_case(k)
    gm = TasuModel(geo, HipOps(), "cuda")
    gm.load_reference_state_dict(sd)
    cm = TasuModel(geo, FakeOps(), "cpu")
    cm.load_reference_state_dict(sd)
    sg, sc = run_text(gm, batch), run_text(cm, batch)
    lg = float(sg.dev["loss_out"][0])
    assert abs(lg - float(z["loss"])) < 2e-2 and abs(lg - float(sc.dev["loss_out"][0])) < 2e-3
    valid = torch.from_numpy(sg.plan.key_mask[:, : sg.S].astype(bool))
    cols = torch.from_numpy(z["cols"])
    ref = torch.from_numpy(z["logits_cols"])
    assert float((gm.logits_view(sg).float().cpu()[:, :, cols] - ref)[valid].abs().max() / ref[valid].abs().max()) < 3e-2
    gg, gc = gm.projector_grads(), cm.projector_grads()
    for name in gc:
        assert cosine(gg[name], gc[name]) > 0.9995, name
    assert cosine(gg["encoder_projector.conv1d.bias"], torch.from_numpy(z["grad.conv1d.bias"])) > 0.995
    assert cosine(gg["encoder_projector.conv1d.weight"][::8], torch.from_numpy(z["grad.conv1d.weight.rows8"])) > 0.995
    assert cosine(gg["encoder_projector.linear1.bias"], torch.from_numpy(z["grad.linear1.bias"])) > 0.995
    assert cosine(gg["encoder_projector.linear1.weight"][::64], torch.from_numpy(z["grad.linear1.weight.rows64"])) > 0.995


def test_cross_attention_projector_on_gpu():
    """``encoder_projector="cross-attention"`` (EncoderProjectorCTCCA) on the HIP kernels (score / value GEMMs per head against
    the embedding table, tasu_scale_softmax_rows_bf16 / tasu_softmax_bwd_rows_bf16 between them): against the REAL reference's
    fp32 golden (|loss - ref| <= 2e-2, logits <= 3 % of their range, W_q gradient cosine >= 0.99) and against the same host
    code on the CPU double (loss 2e-3, gradient cosine 0.999)."""
    from conftest import ca_projector_case
    from ps_slm_amd.ops import HipOps
    geo, sd, batch, z = ca_projector_case()
    gm = TasuModel(geo, HipOps(), "cuda")
    gm.load_reference_state_dict(sd)
    cm = TasuModel(geo, FakeOps(), "cpu")
    cm.load_reference_state_dict(sd)
    sg, sc = run_text(gm, batch), run_text(cm, batch)
    lg = float(sg.dev["loss_out"][0])
    assert abs(lg - float(z["loss"])) < 2e-2 and abs(lg - float(sc.dev["loss_out"][0])) < 2e-3
    valid = torch.from_numpy(sg.plan.key_mask[:, : sg.S].astype(bool))
    cols = torch.from_numpy(z["cols"])
    ref = torch.from_numpy(z["logits_cols"])
    assert float((gm.logits_view(sg).float().cpu()[:, :, cols] - ref)[valid].abs().max() / ref[valid].abs().max()) < 3e-2
    gg, gc = gm.projector_grads(), cm.projector_grads()
    assert cosine(gg["encoder_projector.W_q.weight"], gc["encoder_projector.W_q.weight"]) > 0.999
    assert cosine(gg["encoder_projector.W_q.weight"], torch.from_numpy(z["grad.W_q.weight"])) > 0.99


def test_generate_margin_cases_exact_on_gpu():
    """Token ids are index work: on the 17 rounding-stable decode cases of tests/golden/mid_generate_margin.npz (14 short + 3 with 40-50 generated positions; 1-4 beams,
    min_length, length penalties, left padding, EOS events; oracle/make_golden_generate_margin.py) the HIP decode path must
    EQUAL the REAL reference's generate() tokens -- which the bf16 oracle and the CPU double also reproduce exactly
    (tests/test_oracle_golden.py, tests/test_host_model_cpu.py)."""
    from conftest import decode_margin_cases
    from ps_slm_amd.decode import beam_search_generate
    from ps_slm_amd.ops import HipOps
    geo, sd, cases = decode_margin_cases()
    gm = TasuModel(geo, HipOps(), "cuda")
    gm.load_reference_state_dict(sd)
    bad = []
    for n, c in enumerate(cases):
        st = gm.prepare_text(c["ids"], c["am"], None, c["post_ids"], None, None)
        gm.forward_projector_text(st)
        toks = beam_search_generate(gm, st, eos_token_id=geo.eos_id, pad_token_id=geo.eos_id, **c["kw"]).numpy()
        if toks.shape != c["tokens"].shape or not np.array_equal(toks, c["tokens"]):
            bad.append((n, toks.tolist(), c["tokens"].tolist()))
    assert not bad, bad


def fp32_model(geo, sd):
    from ps_slm_amd.ops import HipOps
    gm = TasuModel(geo, HipOps(), "cuda")
    gm.llm.keep_f32 = True                                    # what model_factory does for train_config.use_fp16 = false
    gm.arith = "fp32"
    gm.load_reference_state_dict(sd)
    return gm


def test_generate_fp32_mode_equals_the_reference_on_unfiltered_cases():
    """train_config.use_fp16 = false (the reference's decode arithmetic, Multitask/inference_batch.py:113-117): the fp32 decode
    path (ps_slm_amd/decode_fp32.py, csrc/fp32.hip) must EQUAL the REAL reference's generate() tokens on all 24 UNFILTERED random
    cases of tests/golden/mid_generate_fp32.npz -- prompts kept whether or not rounding can flip them (8 of them decode
    differently in the bf16-mode oracle), 1-4 beams, min_length, length penalties, left padding, early EOS -- and on the 17
    rounding-stable cases the bf16 path is pinned on.  The bf16 path on the same 24 cases is reported, not asserted."""
    from conftest import decode_fp32_cases, decode_margin_cases
    from ps_slm_amd.decode import beam_search_generate
    from ps_slm_amd.decode_fp32 import beam_search_generate_fp32
    geo, sd, cases, bf16_agrees = decode_fp32_cases()
    gm = fp32_model(geo, sd)
    bad, bf16_same = [], 0
    for n, c in enumerate(cases):
        st = gm.prepare_text(c["ids"], c["am"], None, c["post_ids"], None, None)
        toks = beam_search_generate_fp32(gm, st, eos_token_id=geo.eos_id, pad_token_id=geo.eos_id, **c["kw"]).numpy()
        if toks.shape != c["tokens"].shape or not np.array_equal(toks, c["tokens"]):
            bad.append((n, toks.tolist(), c["tokens"].tolist()))
        st = gm.prepare_text(c["ids"], c["am"], None, c["post_ids"], None, None)
        gm.forward_projector_text(st)
        t16 = beam_search_generate(gm, st, eos_token_id=geo.eos_id, pad_token_id=geo.eos_id, **c["kw"]).numpy()
        bf16_same += int(t16.shape == c["tokens"].shape and np.array_equal(t16, c["tokens"]))
    print(f"fp32 path: {len(cases) - len(bad)} / {len(cases)} cases exact; bf16 path on the same cases: {bf16_same} / {len(cases)} "
          f"(bf16-mode oracle: {sum(bf16_agrees)})")
    assert not bad, bad
    assert bf16_same < len(cases)                              # the set is not a rounding-stable selection
    _, sd_m, margin = decode_margin_cases()
    gm2 = fp32_model(geo, sd_m)
    for n, c in enumerate(margin):
        st = gm2.prepare_text(c["ids"], c["am"], None, c["post_ids"], None, None)
        toks = beam_search_generate_fp32(gm2, st, eos_token_id=geo.eos_id, pad_token_id=geo.eos_id, **c["kw"]).numpy()
        assert toks.shape == c["tokens"].shape and np.array_equal(toks, c["tokens"]), (n, toks, c["tokens"])


def test_generate_fp32_mode_text_and_audio_fixture():
    """mid_generate_beam4.npz (the reference's fp32 tokens of a text and an audio batch): the fp32 path reproduces BOTH exactly --
    the audio branch through the fp32 encoder (encoder_posterior_fp32: fp32 LayerNorms, GEMMs, bidirectional attention, FSMN, CTC
    softmax) and PSD on the fp32 posterior.  (The bf16 path is only asked for a common prefix >= 8 on this fixture.)"""
    from conftest import mid_audio_psd_case, split_flat
    from ps_slm_amd.decode_fp32 import beam_search_generate_fp32
    geo, sd, batch, _ = mid_audio_psd_case()
    z = load_npz("mid_generate_beam4")
    word_ids = split_flat(z["post_ids_flat"], z["post_lens"])
    ids, am = torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"])
    gm = fp32_model(geo, sd)
    assert gm.encoder.keep_f32 and gm.encoder.ctc_f32 is not None
    st = gm.prepare_text(ids, am, None, word_ids, None, None)
    t = beam_search_generate_fp32(gm, st, max_new_tokens=16).numpy()
    assert np.array_equal(t, z["tokens_text"]), (t, z["tokens_text"])
    st = gm.prepare_audio(ids, am, None, batch["input_features"][:2], batch["input_feature_length"][:2])
    a = beam_search_generate_fp32(gm, st, max_new_tokens=16).numpy()
    assert np.array_equal(a, z["tokens_audio"]), (a, z["tokens_audio"])


def test_eval_forward_in_fp32_equals_the_reference_to_fp32_rounding():
    """train_config.use_fp16 = false outside training (the reference's evaluation(), Multitask/utils/deepspeed_utils.py:394-498, runs
    without autocast): ps_slm_amd.decode_fp32.forward_fp32.  Goldens = the REAL reference's fp32 forward
    (tests/golden/mid_text_clean.npz, mid_audio_psd.npz).  Where the bf16 step is held to |dloss| <= 2e-2 and 3 % of the logit range,
    the fp32 forward is held to what summation order leaves: |dloss| <= 2e-5, logits and log-sum-exp within 2e-5 of their scale,
    the same argmax on every position the reference's tie-free rows allow, the same accuracy, and (audio) the same PSD lengths."""
    from conftest import mid_audio_psd_case
    from ps_slm_amd.decode_fp32 import forward_fp32
    geo = Geometry.from_dict(MID_GEOMETRY)
    # text branch
    z = load_npz("mid_text_clean")
    sd = random_state_dict(geo, int(z["seed_w"]), with_encoder=False)
    batch = synthetic_text_batch(geo, 3, seed=int(z["seed_b"]), prompt_len=9, n_audio=21, target_len=17, speech_pos=4,
                                 feat_frames=12, noise=True, drop_prob=0.15, ragged=True)
    batch["post_ids"] = [list(np.asarray(p)[np.asarray(k, dtype=bool)]) for p, k in zip(batch["post_ids"], batch["keeps"])]
    gm = fp32_model(geo, sd)
    st = gm.prepare_text(batch["input_ids"], batch["attention_mask"], batch["labels"], batch["post_ids"], None, None)
    forward_fp32(gm, st)
    torch.cuda.synchronize()

    def check(st, z, argmax=True):
        res = st.dev["loss_out"].cpu()
        assert abs(float(res[0]) - float(z["loss"])) < 2e-5 * max(1.0, abs(float(z["loss"]))), (float(res[0]), float(z["loss"]))
        assert abs(float(res[1]) - float(z["acc"])) < 1e-6
        valid = torch.from_numpy(st.plan.key_mask[:, : st.S].astype(bool))
        lg = gm.logits_view(st).cpu()
        assert lg.dtype == torch.float32
        ref = torch.from_numpy(z["logits_cols"])
        assert float((lg[:, :, torch.from_numpy(z["cols"])] - ref)[valid].abs().max() / ref[valid].abs().max()) < 2e-5
        lse = st.dev["row_lse"].cpu().view(st.B, st.S)
        assert float((lse - torch.from_numpy(z["lse"]))[valid].abs().max()) < 2e-5 * float(np.abs(z["lse"]).max())
        if argmax and "argmax" in z:
            am = st.dev["row_arg"].cpu().view(st.B, st.S).long()
            assert torch.equal(am[valid], torch.from_numpy(z["argmax"]).long()[valid])

    check(st, z)
    # audio branch: fp32 encoder -> CTC softmax -> PSD (merging + blank filter) -> fp32 projector -> decoder
    geo_a, sd_a, batch_a, za = mid_audio_psd_case()
    gm = fp32_model(geo_a, sd_a)
    st = gm.prepare_audio(batch_a["input_ids"], batch_a["attention_mask"], batch_a["labels"], batch_a["input_features"],
                          batch_a["input_feature_length"], fp32=True)
    assert np.array_equal(st.dev["psd_lens"], za["psd_lens"])
    forward_fp32(gm, st)
    torch.cuda.synchronize()
    check(st, za, argmax=False)


def test_training_step_in_fp32_equals_the_reference_to_fp32_rounding():
    """train_config.use_fp16 = false during training (the shipped recipe): ps_slm_amd/train_fp32.py -- fp32 forward with kept
    activations, fp32 dgrad through the frozen decoder (attention probabilities recomputed), projector weight gradients into the
    flat bucket.  Goldens = the REAL reference's fp32 step (tests/golden/mid_text_clean.npz text branch, mid_audio_psd.npz audio branch
    through the fp32 encoder + PSD).  Where the bf16 step is held to |dloss| <= 2e-2 and gradient cosine >= 0.995, the fp32 step is
    held to |dloss| <= 2e-5 and every projector gradient within 2e-4 (relative L2) of the reference's."""
    from conftest import mid_audio_psd_case
    from ps_slm_amd.train_fp32 import forward_train_fp32
    geo = Geometry.from_dict(MID_GEOMETRY)

    def check_grads(gm, z):
        gg = gm.projector_grads()
        seen = 0
        for k, g in gg.items():
            short = "grad." + k[len("encoder_projector."):]
            ref = None
            if short in z:
                ref = torch.from_numpy(z[short])
            elif short + ".even_rows" in z:
                ref, g = torch.from_numpy(z[short + ".even_rows"]), g[::2]
            if ref is None:
                continue
            seen += 1
            err = float((g.cpu().double() - ref.double()).norm() / ref.double().norm())
            assert err < 2e-4, (k, err)
        assert seen >= 4

    z = load_npz("mid_text_clean")
    sd = random_state_dict(geo, int(z["seed_w"]), with_encoder=False)
    batch = synthetic_text_batch(geo, 3, seed=int(z["seed_b"]), prompt_len=9, n_audio=21, target_len=17, speech_pos=4,
                                 feat_frames=12, noise=True, drop_prob=0.15, ragged=True)
    batch["post_ids"] = [list(np.asarray(p)[np.asarray(k, dtype=bool)]) for p, k in zip(batch["post_ids"], batch["keeps"])]
    del batch["alphas"], batch["keeps"]
    gm = fp32_model(geo, sd)
    st = gm.prepare_text(batch["input_ids"], batch["attention_mask"], batch["labels"], batch["post_ids"], None, None)
    forward_train_fp32(gm, st)
    gm.run_backward(st)
    torch.cuda.synchronize()
    res = st.dev["loss_out"].cpu()
    assert abs(float(res[0]) - float(z["loss"])) < 2e-5 * max(1.0, abs(float(z["loss"]))) and abs(float(res[1]) - float(z["acc"])) < 1e-6
    check_grads(gm, z)
    # the same bits on a second run (deterministic sums), and the bf16 step of the same model is the looser neighbour it should be
    g1 = gm.proj.g.clone()
    st = gm.prepare_text(batch["input_ids"], batch["attention_mask"], batch["labels"], batch["post_ids"], None, None)
    forward_train_fp32(gm, st)
    gm.run_backward(st)
    torch.cuda.synchronize()
    assert torch.equal(gm.proj.g, g1)
    st16 = run_text(gm, batch)
    assert 1e-6 < abs(float(st16.dev["loss_out"][0]) - float(z["loss"])) < 2e-2
    # audio branch
    geo_a, sd_a, batch_a, za = mid_audio_psd_case()
    gm = fp32_model(geo_a, sd_a)
    st = gm.prepare_audio(batch_a["input_ids"], batch_a["attention_mask"], batch_a["labels"], batch_a["input_features"],
                          batch_a["input_feature_length"], fp32=True)
    forward_train_fp32(gm, st)
    gm.run_backward(st)
    torch.cuda.synchronize()
    assert abs(float(st.dev["loss_out"][0]) - float(za["loss"])) < 2e-5 * max(1.0, abs(float(za["loss"])))
    check_grads(gm, za)


def test_use_fp16_false_selects_the_fp32_decode_through_the_plugin():
    """The reference's own flag picks the arithmetic (Multitask/scripts/decode_sensevoice.sh runs with use_fp16 unset = false):
    ``model_factory(train_config.use_fp16=false)`` keeps fp32 copies of the frozen weights and ``model.generate`` runs the fp32
    path -- the same tokens as calling it directly, and the bf16 path is what ``use_fp16=true`` gets."""
    from ps_slm_amd.config import ModelConfig, TrainConfig
    from ps_slm_amd.decode_fp32 import beam_search_generate_fp32
    from ps_slm_amd.ps_slm import model_factory
    outs = {}
    for fp16 in (False, True):
        tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, gt_emb_noise=False, ctc_posterior=True, do_psd=True, use_fp16=fp16)
        mc = ModelConfig(llm_path="synthetic:mid", encoder_projector="linear-silu", llm_dim=256)
        model, tok = model_factory(tc, mc, device="cuda:0", init_seed=77)
        core = model.core
        assert core.arith == ("bf16" if fp16 else "fp32") and (core.llm.f32 is None) == fp16
        raw = synthetic_text_batch(core.geo, 2, seed=5, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=8, noise=False)
        ids = raw["input_ids"][:, :10]
        am = torch.ones_like(ids, dtype=torch.bool)
        targets = ["ab cde f ghij kl m", "no pq rst uvw"]       # (generate() strips everything but letters: ps-slm.py:592-596)
        model.eval()
        outs[fp16] = model.generate(input_ids=ids, attention_mask=am, targets=targets, num_beams=4, max_new_tokens=12).numpy()
        if not fp16:
            # ... and an eval-mode model(**batch) runs the fp32 forward (fp32 logits, no autograd node)
            call = dict(input_ids=raw["input_ids"], attention_mask=raw["attention_mask"], labels=raw["labels"], input_features=None,
                        input_feature_length=None, GT=[" ".join(map(str, p)) for p in raw["post_ids"]])
            out, acc = model(**call)
            assert out.logits.dtype == torch.float32 and not out.loss.requires_grad and 0.0 < float(out.loss.detach()) < 20.0
            model.train()
            out_t, _ = model(**call)                            # training mode: the fp32 step, behind the same autograd node
            assert out_t.loss.requires_grad and model.last_state.fp32
            assert abs(float(out_t.loss.detach()) - float(out.loss.detach())) < 1e-5 * float(out.loss.detach())
            out_t.loss.backward()
            assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
            model.eval()
            ids_list = [model.encoder_tokenizer.encode(t) for t in targets]
            assert all(len(p) > 0 for p in ids_list)
            st = core.prepare_text(ids, am, None, ids_list, None, None)
            direct = beam_search_generate_fp32(core, st, num_beams=4, max_new_tokens=12, eos_token_id=tok.eos_token_id,
                                               pad_token_id=tok.pad_token_id).numpy()
            assert np.array_equal(outs[False], direct)
    assert outs[True].shape[0] == outs[False].shape[0] == 2


@pytest.mark.parametrize("nb,min_len,lpw", [(2, 6, 2.0), (3, 4, 0.5), (1, 1, 1.0)])
def test_generate_other_settings_on_gpu(setup, nb, min_len, lpw):
    """Other beam counts, a minimum length (EOS banned in the log-prob/top-k kernel for the first positions) and length
    penalties: GPU decode vs the same host code on the CPU double (equal on most rows; a row that splits before position 6 must
    split at a near-tie of the double's candidates: log-probs within 0.07)."""
    from ps_slm_amd.decode import beam_search_generate
    geo, sd, gm, cm = setup
    batch = synthetic_text_batch(geo, 4, seed=17, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=12, noise=False)
    ids, am = batch["input_ids"][:, :9], batch["attention_mask"][:, :9]
    outs, rec = [], []
    orig = cm.ops.beam_update

    def hook(vals, idx, bs, first):                               # the double's candidates (log-prob, token) of every step
        rec.append((vals.clone().numpy(), idx.clone().numpy()))
        return orig(vals, idx, bs, first)
    cm.ops.beam_update = hook
    try:
        for m in (gm, cm):
            st = m.prepare_text(ids, am, None, batch["post_ids"], None, None)
            m.forward_projector_text(st)
            outs.append(beam_search_generate(m, st, num_beams=nb, max_new_tokens=12, min_length=min_len, length_penalty=lpw).numpy())
    finally:
        cm.ops.beam_update = orig
    g, c = outs
    n = min(g.shape[1], c.shape[1])
    assert n >= min_len
    common = (g[:, :n] == c[:, :n]).cumprod(1).sum(1)
    assert (common == n).mean() >= 0.5, (g, c)
    for r in np.nonzero(common < min(6, n))[0]:
        # an early split must be a NEAR-TIE of the double's own candidates (a random-init model is full of them, and the GPU path
        # rounds the normed activations at a different point: tasu_gemm_stream_resid_prenorm), not a different search.  Greedy
        # decode only: with beams the candidate rows are re-ordered per step.
        assert nb == 1, (g, c)
        vals, idx = rec[int(common[r])]
        j = np.nonzero(idx[r] == g[r, common[r]])[0]
        assert len(j) == 1 and abs(float(vals[r, 0] - vals[r, j[0]])) < 0.07, (r, g[r], c[r], vals[r], idx[r])
    assert not (g[:, : min_len - 1] == geo.eos_id).any()          # EOS cannot appear before min_length


def test_generate_more_than_64_beams_rows(setup):
    """20 utterances x 4 beams = 80 rows: the weight-streaming kernels run in two row chunks (64 + 16).  Row arithmetic does
    not depend on the batch a row sits in, so the tokens must be EXACTLY those of the same utterances decoded ten at a time
    (one chunk); against the CPU double most rows agree (a random-init model is full of near-ties that bf16 flips)."""
    from ps_slm_amd.decode import beam_search_generate
    geo, sd, gm, cm = setup
    batch = synthetic_text_batch(geo, 20, seed=91, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=12, noise=False)
    ids, am = batch["input_ids"][:, :9], batch["attention_mask"][:, :9]

    def decode(m, rows):
        st = m.prepare_text(ids[rows], am[rows], None, [batch["post_ids"][r] for r in rows], None, None)
        m.forward_projector_text(st)
        return beam_search_generate(m, st, max_new_tokens=12).numpy()
    g_all = decode(gm, list(range(20)))
    g_lo, g_hi = decode(gm, list(range(10))), decode(gm, list(range(10, 20)))
    n = min(g_all.shape[1], g_lo.shape[1], g_hi.shape[1])
    assert np.array_equal(g_all[:10, :n], g_lo[:, :n]) and np.array_equal(g_all[10:, :n], g_hi[:, :n])
    c = decode(cm, list(range(20)))
    n = min(n, c.shape[1])
    assert ((g_all[:, :n] == c[:, :n]).cumprod(1).sum(1) >= 8).mean() >= 0.7


def test_graph_replay_matches_eager(setup):
    """hipGraph capture/replay of the forward and backward launch sequences gives bit-identical results."""
    geo, sd, gm, _ = setup
    batch = synthetic_text_batch(geo, 3, seed=77, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=12,
                                 noise=False)
    def one(model):
        st = model.prepare_text(batch["input_ids"], batch["attention_mask"], batch["labels"], batch["post_ids"], None, None)
        model.run_forward_text(st)
        model.run_backward(st)
        torch.cuda.synchronize()
        return st.dev["loss_out"].clone(), model.proj.g.clone()
    gm.use_graphs = False
    l0, g0 = one(gm)
    gm.use_graphs = True
    outs = [one(gm) for _ in range(4)]          # eager warm-up, capture + replay, replay, replay
    assert len(gm._graphs) == 2
    for l, g in outs:
        assert torch.equal(l, l0) and torch.equal(g, g0)
    # a LARGER batch grows the named workspace buffers: the graphs above hold freed addresses and must not be replayed
    big = synthetic_text_batch(geo, 7, seed=78, prompt_len=9, n_audio=40, target_len=30, speech_pos=4, feat_frames=12, noise=False)
    stb = gm.prepare_text(big["input_ids"], big["attention_mask"], big["labels"], big["post_ids"], None, None)
    gm.run_forward_text(stb)
    gm.run_backward(stb)
    torch.cuda.synchronize()
    again = [one(gm) for _ in range(3)]          # eager, re-capture, replay -- all on the new buffers
    gm.use_graphs = False
    for l, g in again:
        assert torch.equal(l, l0) and torch.equal(g, g0)
    gm._graphs.clear()
    gm._graph_seen.clear()


def test_bucketed_shapes_reuse_graphs(setup):
    """Real data gives every batch its own shape; with TasuModel.shape_buckets three batches of different raw shapes fall into
    one bucket, so the graphs captured on the second are replayed for the third -- and every bucketed, graph-replayed step
    matches the eager step on the unpadded batch (loss to summation order, gradients to bf16 rounding of the padded GEMMs)."""
    from ps_slm_amd.ops import HipOps
    geo, sd, gm, _ = setup
    bm = TasuModel(geo, HipOps(), "cuda")
    bm.load_reference_state_dict(sd)
    bm.use_graphs, bm.shape_buckets = True, (16, 8, 256)
    gm.use_graphs = False
    shapes = ((3, 21, 17), (4, 19, 17), (5, 20, 18), (6, 21, 16))
    for i, (seed, n_audio, tl) in enumerate(shapes):
        batch = synthetic_text_batch(geo, 3, seed=seed, prompt_len=9, n_audio=n_audio, target_len=tl, speech_pos=4, feat_frames=12,
                                     noise=True, drop_prob=0.0, ragged=True)
        outs = []
        for m in (gm, bm):
            st = m.prepare_text(batch["input_ids"], batch["attention_mask"], batch["labels"], batch["post_ids"],
                                batch.get("alphas"), batch.get("keeps"))
            m.run_forward_text(st)
            m.run_backward(st)
            torch.cuda.synchronize()
            outs.append((st.dev["loss_out"].clone(), m.proj.g.clone(), st.S))
        (l0, g0, s0), (l1, g1, s1) = outs
        assert s1 % 16 == 0 and 0 <= s1 - s0 < 16
        assert abs(float(l0[0]) - float(l1[0])) < 1e-5 * float(l0[0]) + 1e-6 and float(l0[2]) == float(l1[2])
        assert float((g0 - g1).norm() / g0.norm()) < 5e-3
        assert len(bm._graphs) == (0 if i == 0 else 2)          # eager, captured, then replayed: one forward + one backward graph


def test_encoder_graph_replay_matches_eager():
    """The SANM encoder's launch sequence replayed as a hipGraph: bit-identical CTC posterior, PSD lengths and loss for
    inputs that CHANGE between the capture and the replays (the uploads stay outside the captured region)."""
    from conftest import mid_audio_psd_case
    from ps_slm_amd.ops import HipOps
    geo, sd, batch, _ = mid_audio_psd_case()
    gm = TasuModel(geo, HipOps(), "cuda")
    gm.load_reference_state_dict(sd)
    g = torch.Generator().manual_seed(5)
    variants = [dict(batch, input_features=batch["input_features"] + 0.3 * i * torch.randn(batch["input_features"].shape, generator=g))
                for i in range(4)]

    def one(b):
        st = gm.prepare_audio(b["input_ids"], b["attention_mask"], b["labels"], b["input_features"], b["input_feature_length"])
        gm.run_forward_llm(st)
        gm.run_backward(st)
        torch.cuda.synchronize()
        Te = b["input_features"].shape[1] + 4
        # (the step's audio front end works from the CTC head's bf16 logits; the fp32 posterior is not materialised)
        post = gm._buf("enc_ctc_logits", (b["input_features"].shape[0] * Te, rup64(geo.ctc_vocab)), torch.bfloat16)[:, : geo.ctc_vocab].clone()
        return post, np.array(st.dev["psd_lens"]).copy(), st.dev["loss_out"].clone()

    def rup64(v):
        return (v + 63) // 64 * 64

    gm.use_graphs = False
    want = [one(b) for b in variants]
    gm.use_graphs = True
    got = [one(b) for b in variants]                  # eager warm-up, capture + replay, replay, replay
    assert any(k[0] == "region" and k[1] == "encoder" for k in gm._graphs)
    for (p0, n0, l0), (p1, n1, l1) in zip(want, got):
        assert torch.equal(p0, p1) and np.array_equal(n0, n1) and torch.equal(l0, l1)
    assert not torch.equal(want[0][0], want[3][0])     # the variants really differ


@pytest.mark.parametrize("graphs", [False, True])
def test_encoder_one_batch_ahead_changes_nothing(graphs):
    """TasuModel.prefetch_encoder: the frozen encoder pass of batch i + 1 on a side stream under batch i's decoder step.  Same
    kernels on the same data: logits, PSD lengths, loss and gradients are bit-identical to the in-line order -- for announced
    batches, for a batch that was NOT the announced one (falls back in line, after the pending pass), and after a decode call."""
    from conftest import mid_audio_psd_case
    from ps_slm_amd.ops import HipOps
    geo, sd, batch, _ = mid_audio_psd_case()
    g = torch.Generator().manual_seed(11)
    variants = [dict(batch, input_features=batch["input_features"] + 0.3 * i * torch.randn(batch["input_features"].shape, generator=g))
                for i in range(5)]
    order = [0, 1, 2, 3, 4, 1, 3]
    if not graphs:
        # eager launches prefetch ANY shape (dynamic batching): a batch of twice the frames makes the encoder's workspace grow
        # inside a prefetched pass, a batch of two utterances follows it
        f5 = torch.cat([variants[1]["input_features"], variants[2]["input_features"]], 1)
        variants.append(dict(batch, input_features=f5, input_feature_length=batch["input_feature_length"] + f5.shape[1] // 2))
        variants.append({k: (v[:2] if isinstance(v, torch.Tensor) else v) for k, v in variants[3].items()})
        order = [0, 1, 5, 2, 4, 1, 6, 5, 3]
    zd = load_npz("mid_generate_beam4")
    dec_ids, dec_am = torch.from_numpy(zd["input_ids"]), torch.from_numpy(zd["attention_mask"])

    def run(ahead):
        gm = TasuModel(geo, HipOps(), "cuda")
        gm.load_reference_state_dict(sd)
        gm.use_graphs = graphs
        outs, hits = [], 0
        for n, i in enumerate(order):
            b = variants[i]
            st = gm.prepare_audio(b["input_ids"], b["attention_mask"], b["labels"], b["input_features"], b["input_feature_length"])
            gm.run_forward_llm(st)
            if ahead and n + 1 < len(order):
                # step 4 announces the WRONG batch (variant 0 instead of 1); step 5 announces the right one but a decode call on
                # another batch comes in between and takes over the encoder's workspace
                nb = variants[0] if n == 4 else variants[order[n + 1]]
                hits += bool(gm.prefetch_encoder(nb["input_features"], nb["input_feature_length"]))
            gm.run_backward(st)
            if n == 5:                                 # an inference front end (other batch, B = 2) between two training steps
                gm.prepare_audio(dec_ids, dec_am, None, variants[2]["input_features"][:2], batch["input_feature_length"][:2])
            torch.cuda.synchronize()
            outs.append((np.array(st.dev["psd_lens"]).copy(), st.dev["loss_out"].clone(), gm.proj.g.clone()))
        return outs, hits

    want, _ = run(False)
    got, hits = run(True)
    assert hits >= (3 if graphs else 4)                # the first passes of a shape allocate (and capture) in line
    for (n0, l0, g0), (n1, l1, g1) in zip(want, got):
        assert np.array_equal(n0, n1) and torch.equal(l0, l1) and torch.equal(g0, g1)
    assert not torch.equal(want[0][1], want[1][1])
