"""Pins oracle/tasu_oracle.py against fixtures produced from the REAL reference (oracle/make_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import golden_batch, load_npz
from oracle import tasu_oracle as O

FP32_TOL = dict(rtol=2e-4, atol=2e-5)


def close(a, b, **tol):
    a = (a.detach() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))).float()
    b = (b.detach() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b))).float()
    assert a.shape == b.shape, (a.shape, b.shape)
    torch.testing.assert_close(a, b, **tol)


@pytest.mark.parametrize("case", ["text_clean_right", "text_noise_right", "text_clean_left", "text_clean_b1"])
def test_text_forward_backward_fp32(case, geo, tiny_weights):
    b, z = golden_batch(case)
    out, grads = O.loss_and_projector_grads(tiny_weights, b, geo, "fp32")
    close(out["loss"], z["loss"], rtol=1e-5, atol=1e-6)
    valid = out["mask"]
    close(out["logits"][valid], torch.from_numpy(z["logits"])[valid], **FP32_TOL)
    close(out["acc"], z["acc"], rtol=0, atol=1e-7)
    for k, g in grads.items():
        close(g, z["grad." + k[len("encoder_projector."):]], rtol=2e-4, atol=1e-6)


def test_text_forward_backward_random_batches_vs_reference(geo, tiny_weights):
    """8 random text-only batches through the REAL reference's forward + backward (oracle/make_golden_text_random.py)."""
    from conftest import split_flat
    z = load_npz("text_random")
    for n in range(int(z["n_cases"])):
        b = {k: torch.from_numpy(z[f"c{n}_{k}"]) for k in ("input_ids", "attention_mask", "labels")}
        b["post_ids"] = split_flat(z[f"c{n}_post_ids_flat"], z[f"c{n}_post_lens"])
        out, grads = O.loss_and_projector_grads(tiny_weights, b, geo, "fp32")
        close(out["loss"], z[f"c{n}_loss"], rtol=1e-5, atol=1e-6)
        close(out["acc"], z[f"c{n}_acc"], rtol=0, atol=1e-7)
        valid = out["mask"]
        close(out["logits"][valid], torch.from_numpy(z[f"c{n}_logits"])[valid], **FP32_TOL)
        for k, g in grads.items():
            close(g, z[f"c{n}_grad." + k[len("encoder_projector."):]], rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize("case", ["text_clean_right", "text_clean_left"])
def test_merge_plan(case, geo, tiny_weights):
    b, z = golden_batch(case)
    plens = torch.tensor([len(p) for p in b["post_ids"]])
    plan = O.merge_plan(b["input_ids"], b["attention_mask"], plens, geo["speech_id"])
    assert np.array_equal(plan["mask"].numpy(), z["merged_mask"].astype(bool))
    assert np.array_equal(plan["position_ids"].numpy(), z["merged_position_ids"])
    out = O.forward_text(tiny_weights, b, geo)
    assert np.array_equal(out["labels"].numpy(), z["merged_labels"])
    if "merged_embeds" in z:
        close(out["embeds"], z["merged_embeds"], **FP32_TOL)
        close(out["proj"], z["proj_out"], **FP32_TOL)


def test_merge_rejects_two_sided_padding(geo):
    ids = torch.tensor([[1, geo["speech_id"], 2], [3, geo["speech_id"], 4]])
    am = torch.tensor([[0, 1, 1], [1, 1, 0]]).bool()
    with pytest.raises(ValueError):
        O.merge_plan(ids, am, torch.tensor([2, 2]), geo["speech_id"])


def test_bf16_mode_vs_reference_autocast(geo, tiny_weights):
    """Oracle bf16 emulation vs the reference's pieces under torch.autocast('cpu', bfloat16)."""
    b, _ = golden_batch("text_clean_right")
    z = load_npz("text_clean_right_bf16")
    out, grads = O.loss_and_projector_grads(tiny_weights, b, geo, "bf16")
    close(out["proj"], z["proj_out"], rtol=2e-2, atol=2e-3)
    valid = out["mask"]
    ref = torch.from_numpy(z["logits"])
    err = (out["logits"] - ref)[valid].abs().max() / ref[valid].abs().max()
    assert err < 3e-2, err
    assert abs(float(out["loss"]) - float(z["loss"])) < 2e-2
    for k, g in grads.items():
        r = torch.from_numpy(z["grad." + k[len("encoder_projector."):]])
        cos = torch.nn.functional.cosine_similarity(g.flatten(), r.flatten(), dim=0)
        assert cos > 0.995, (k, cos)


def test_encoder_ragged(geo, tiny_weights):
    z = load_npz("encoder_ragged")
    enc, olens = O.sensevoice_encoder(tiny_weights, torch.from_numpy(z["speech"]), torch.from_numpy(z["speech_lengths"]),
                                      geo["enc_heads"], geo["enc_kernel"])
    assert np.array_equal(olens.numpy(), z["olens"])
    close(enc, z["enc_out"], **FP32_TOL)
    post = torch.softmax(O.linear(enc, tiny_weights["encoder.ctc.ctc_lo.weight"],
                                  tiny_weights["encoder.ctc.ctc_lo.bias"], "fp32"), -1)
    close(post, z["ctc_posterior"], **FP32_TOL)


def test_encoder_random_batches_vs_reference(geo, tiny_weights):
    """8 random ragged batches through the REAL reference's SANM encoder + CTC head (oracle/make_golden_encoder.py): frame counts
    below, at and above the FSMN kernel width.  Rows beyond an utterance's length are padding in both and are not compared."""
    z = load_npz("encoder_random")
    for n in range(int(z["n_cases"])):
        speech, slen = torch.from_numpy(z[f"c{n}_speech"]), torch.from_numpy(z[f"c{n}_speech_lengths"])
        enc, olens = O.sensevoice_encoder(tiny_weights, speech, slen, geo["enc_heads"], geo["enc_kernel"])
        assert np.array_equal(olens.numpy(), z[f"c{n}_olens"]), n
        post = torch.softmax(O.linear(enc, tiny_weights["encoder.ctc.ctc_lo.weight"], tiny_weights["encoder.ctc.ctc_lo.bias"], "fp32"), -1)
        for b in range(speech.shape[0]):
            L = int(olens[b])
            close(enc[b, :L], z[f"c{n}_enc_out"][b, :L], **FP32_TOL)
            close(post[b, :L], z[f"c{n}_ctc_posterior"][b, :L], **FP32_TOL)


def test_psd_crafted():
    z = load_npz("psd_crafted")
    post = torch.from_numpy(z["posterior"])
    out, nl = O.psd(post, torch.from_numpy(z["lens"]), post, 0)
    assert np.array_equal(nl.numpy(), z["new_lens"])
    close(out, z["out"], rtol=1e-6, atol=1e-7)
    z2 = load_npz("psd_crafted_logprob")
    lp = post.clamp_min(1e-30).log()
    out2, nl2 = O.psd(lp, torch.from_numpy(z["lens"]), lp, 0)
    assert np.array_equal(nl2.numpy(), z2["new_lens"])
    close(out2, z2["out"], rtol=1e-6, atol=1e-7)


def test_psd_empty_batch():
    post = torch.zeros(2, 4, 5)
    post[..., 0] = 1.0
    out, nl = O.psd(post, torch.tensor([4, 0]), post, 0)
    assert out.shape == (2, 0, 5) and nl.tolist() == [0, 0]


def test_audio_path(geo, tiny_weights):
    b, z = golden_batch("audio_psd_right")
    W = dict(tiny_weights)
    W["encoder.ctc.ctc_lo.bias"] = torch.from_numpy(z["encoder.ctc.ctc_lo.bias"])
    W["encoder.ctc.ctc_lo.weight"] = torch.from_numpy(z["encoder.ctc.ctc_lo.weight"])
    post, _, lens = O.audio_front(W, b["input_features"], b["input_feature_length"], geo["enc_heads"], geo["enc_kernel"])
    po, pl = O.psd(post, lens, post, 0)
    assert np.array_equal(pl.numpy(), z["psd_lens"])
    assert pl.max() < lens.max(), "fixture must exercise merging/filtering"
    close(po, z["psd_out"], **FP32_TOL)
    out, grads = O.loss_and_projector_grads(W, b, geo, "fp32", audio=True)
    close(out["loss"], z["loss"], rtol=1e-5, atol=1e-6)
    close(out["logits"][out["mask"]], torch.from_numpy(z["logits"])[out["mask"]], **FP32_TOL)
    for k, g in grads.items():
        close(g, z["grad." + k[len("encoder_projector."):]], rtol=2e-4, atol=1e-6)


def test_adamw_matches_torch():
    torch.manual_seed(0)
    p0 = torch.randn(257)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=5e-5, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0)
    p, m, v = p0.clone(), torch.zeros(257), torch.zeros(257)
    for step in range(1, 6):
        g = torch.randn(257)
        ref.grad = g.clone()
        opt.step()
        O.adamw_step(p, g, m, v, step, 5e-5)
        torch.testing.assert_close(p, ref.data, rtol=1e-6, atol=1e-8)
    # weight decay knob (torch default 0.01; DeepSpeed FusedAdam default 0.0)
    ref2 = torch.nn.Parameter(p0.clone())
    opt2 = torch.optim.AdamW([ref2], lr=1e-3, eps=1e-6, weight_decay=0.01)
    p, m, v = p0.clone(), torch.zeros(257), torch.zeros(257)
    g = torch.randn(257)
    ref2.grad = g.clone()
    opt2.step()
    O.adamw_step(p, g, m, v, 1, 1e-3, weight_decay=0.01)
    torch.testing.assert_close(p, ref2.data, rtol=1e-6, atol=1e-8)


def test_warmup_cosine_shape():
    lr = [O.lr_for_optimizer_step(k, 5e-5) for k in range(1, 15002)]
    assert lr[0] == 0.0 and lr[1] == 0.0  # DeepSpeed: optimizer.step() precedes scheduler.step(); log(1)=0
    assert all(b >= a for a, b in zip(lr[:200], lr[1:201]))
    assert abs(max(lr) - 5e-5) < 1e-9
    assert abs(lr[-1] - 5e-5 * 1e-4) < 1e-9


# ------------------------------------------------------------------ kernel-compatible ("mid") geometry
def mid_setup(name):
    import dataclasses

    from ps_slm_amd.model import Geometry
    from ps_slm_amd.synthetic import MID_GEOMETRY, random_state_dict, synthetic_text_batch

    z = load_npz(name)
    geo = Geometry.from_dict(MID_GEOMETRY)
    sd = random_state_dict(geo, int(z["seed_w"]), with_encoder=True)
    batch = synthetic_text_batch(geo, 3, seed=int(z["seed_b"]), prompt_len=9, n_audio=21, target_len=17, speech_pos=4,
                                 feat_frames=12, noise=True, drop_prob=0.15, ragged=True)
    # the fixture used the clean posterior of the kept ids (see oracle/make_golden.py:main_mid)
    batch["post_ids"] = [list(np.asarray(p)[np.asarray(k, dtype=bool)]) for p, k in zip(batch["post_ids"], batch["keeps"])]
    del batch["alphas"], batch["keeps"]
    return geo, dataclasses.asdict(geo), sd, batch, z


def check_mid(out, grads, z, tol):
    close(out["loss"], z["loss"], rtol=tol, atol=tol)
    lg = out["logits"].detach()
    valid = out["mask"]
    cols = torch.from_numpy(z["cols"])
    close(lg[:, :, cols][valid], torch.from_numpy(z["logits_cols"])[valid], rtol=20 * tol, atol=20 * tol)
    close(torch.logsumexp(lg, -1)[valid], torch.from_numpy(z["lse"])[valid], rtol=20 * tol, atol=20 * tol)
    for k, g in grads.items():
        short = k[len("encoder_projector."):]
        if "grad." + short in z:
            close(g, z["grad." + short], rtol=50 * tol, atol=tol)
        elif "grad." + short + ".even_rows" in z:
            close(g[::2], z["grad." + short + ".even_rows"], rtol=50 * tol, atol=tol)


def test_mid_text_fp32():
    geo, gd, sd, batch, z = mid_setup("mid_text_clean")
    out, grads = O.loss_and_projector_grads(sd, batch, gd, "fp32")
    check_mid(out, grads, z, 1e-5)


def test_mid_audio_fp32():
    geo, gd, sd, batch, z = mid_setup("mid_audio")
    out, grads = O.loss_and_projector_grads(sd, batch, gd, "fp32", audio=True)
    check_mid(out, grads, z, 1e-5)


def test_mid_audio_psd_fp32():
    import dataclasses
    from conftest import mid_audio_psd_case
    geo, sd, batch, z = mid_audio_psd_case()
    gd = dataclasses.asdict(geo)
    post, _, lens = O.audio_front(sd, batch["input_features"], batch["input_feature_length"], geo.enc_heads, geo.enc_kernel)
    _, pl = O.psd(post, lens, post, 0)
    assert np.array_equal(pl.numpy(), z["psd_lens"]) and int(pl.sum()) < int(lens.sum())
    out, grads = O.loss_and_projector_grads(sd, batch, gd, "fp32", audio=True)
    check_mid(out, grads, z, 1e-5)


def test_generate_beam4_text_and_audio(geo, tiny_weights):
    """Beam-4 token ids of the REAL reference's generate() (HF beam search) at the tiny geometry."""
    from conftest import split_flat
    # text path (gt_emb: clean posterior of the regex-cleaned targets)
    z = load_npz("generate_text_beam4")
    ids, am = torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"])
    post_ids = split_flat(z["post_ids_flat"], z["post_lens"])
    post, plen = O.pseudo_posterior(post_ids, geo["ctc_vocab"])
    proj = O.projector(tiny_weights, post)
    emb, mask, _, _ = O.merge(proj, plen, tiny_weights["llm.model.embed_tokens.weight"][ids], ids, am, None, geo["speech_id"])
    toks = O.beam_search_generate(tiny_weights, emb, mask, geo, max_new_tokens=12)
    assert np.array_equal(toks.numpy(), z["tokens"]), (toks, z["tokens"])
    # audio path
    z = load_npz("generate_audio_beam4")
    W = dict(tiny_weights)
    W["encoder.ctc.ctc_lo.bias"] = torch.from_numpy(z["encoder.ctc.ctc_lo.bias"])
    W["encoder.ctc.ctc_lo.weight"] = torch.from_numpy(z["encoder.ctc.ctc_lo.weight"])
    ids, am = torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"])
    post, _, lens = O.audio_front(W, torch.from_numpy(z["input_features"]), torch.from_numpy(z["input_feature_length"]),
                                  geo["enc_heads"], geo["enc_kernel"])
    po, pl = O.psd(post, lens, post, 0)
    proj = O.projector(W, po)
    emb, mask, _, _ = O.merge(proj, pl, W["llm.model.embed_tokens.weight"][ids], ids, am, None, geo["speech_id"])
    toks = O.beam_search_generate(W, emb, mask, geo, max_new_tokens=12)
    assert np.array_equal(toks.numpy(), z["tokens"]), (toks, z["tokens"])


def test_merge_plan_random_batches_vs_reference(geo):
    """48 random ragged batches (right / left padding, with and without labels, 1-5 rows) through the REAL reference's
    _merge_input_ids_with_audio_features (oracle/make_golden_merge.py): the product's host plan (ps_slm_amd.merge) and the
    oracle's restatement must place every token, audio frame and pad exactly where the reference did."""
    from conftest import load_npz
    from ps_slm_amd.merge import build_merge_plan
    z = load_npz("merge_random")
    for n in range(int(z["n_cases"])):
        ids, am, na = z[f"c{n}_input_ids"], z[f"c{n}_attention_mask"], z[f"c{n}_num_audio"]
        labels = z.get(f"c{n}_labels")
        Lmax = int(na.max())
        plan = build_merge_plan(ids, am, labels, na, geo["speech_id"], Lmax)
        B, S = plan.B, plan.S
        tag, mask, pos = z[f"c{n}_source_tag"], z[f"c{n}_mask"].astype(bool), z[f"c{n}_position_ids"]
        assert tag.shape == (B, S), n
        kind, idx = plan.src_kind.reshape(B, S), plan.src_idx.reshape(B, S)
        # decode the reference's tags: token t -> -(t + 1); audio frame (b, j) -> 1000 (b + 1) + j; pad -> 0
        want_kind = np.where(tag < 0, 1, np.where(tag > 0, 2, 0))
        assert np.array_equal(kind, want_kind), n
        assert np.array_equal(idx[kind == 1], (-tag[kind == 1] - 1).astype(np.int64)), n
        bb, jj = np.divmod(tag[kind == 2].astype(np.int64), 1000)
        assert np.array_equal(idx[kind == 2], (bb - 1) * Lmax + jj), n
        assert np.array_equal(np.nonzero(kind == 2)[0], bb - 1), n            # audio rows stay in their own utterance
        assert np.array_equal(plan.key_mask[:, :S].astype(bool), mask), n
        assert np.array_equal(plan.position_ids.reshape(B, S), pos), n
        if labels is not None:
            assert np.array_equal(plan.labels, z[f"c{n}_merged_labels"]), n
        op = O.merge_plan(torch.from_numpy(ids), torch.from_numpy(am), torch.from_numpy(na), geo["speech_id"])
        assert np.array_equal(op["mask"].numpy(), mask) and np.array_equal(op["position_ids"].numpy(), pos), n


def test_psd_random_batches_vs_reference():
    """32 random batches of peaky posteriors through the REAL reference's psd() (oracle/make_golden_psd.py)."""
    z = load_npz("psd_random")
    for n in range(int(z["n_cases"])):
        post, lens = torch.from_numpy(z[f"c{n}_posterior"]), torch.from_numpy(z[f"c{n}_lens"])
        out, nl = O.psd(post, lens, post, 0)
        assert np.array_equal(nl.numpy(), z[f"c{n}_new_lens"]), n
        assert tuple(out.shape) == z[f"c{n}_out"].shape, n
        close(out, z[f"c{n}_out"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("kv_cache", [False, True])
def test_generate_random_cases_vs_reference(geo, tiny_weights, kv_cache):
    """14 decode cases of the REAL reference's generate() (oracle/make_golden_generate.py): 1-3 left-padded utterances,
    1-4 beams, several max_new_tokens / min_length / length_penalty settings -- token ids must be identical, for the oracle's
    whole-sequence form and for its KV-cache form (``qwen2_hidden_step`` on a cache that follows the beams: what HF generate runs,
    and what bench.py's CPU decode baseline times)."""
    from conftest import split_flat
    z = load_npz("generate_random")
    for n in range(int(z["n_cases"])):
        ids, am = torch.from_numpy(z[f"c{n}_input_ids"]), torch.from_numpy(z[f"c{n}_attention_mask"])
        post_ids = split_flat(z[f"c{n}_post_ids_flat"], z[f"c{n}_post_lens"])
        nb, new, min_len = (int(v) for v in z[f"c{n}_kw"])
        post, plen = O.pseudo_posterior(post_ids, geo["ctc_vocab"])
        proj = O.projector(tiny_weights, post)
        emb, mask, _, _ = O.merge(proj, plen, tiny_weights["llm.model.embed_tokens.weight"][ids], ids, am, None, geo["speech_id"])
        toks = O.beam_search_generate(tiny_weights, emb, mask, geo, num_beams=nb, max_new_tokens=new, min_length=min_len,
                                      length_penalty=float(z[f"c{n}_length_penalty"]), kv_cache=kv_cache)
        assert np.array_equal(toks.numpy(), z[f"c{n}_tokens"]), (n, toks, z[f"c{n}_tokens"])


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_generate_margin_cases_vs_reference(mode):
    """14 rounding-stable decode cases at the mid geometry (oracle/make_golden_generate_margin.py): the oracle's beam search
    must reproduce the REAL reference's generate() token ids exactly, in fp32 and in bf16-emulation mode (these are the cases
    the HIP decode path is compared with token by token in tests/test_gpu_model.py)."""
    import dataclasses

    from conftest import decode_margin_cases
    geo, sd, cases = decode_margin_cases()
    gd = dataclasses.asdict(geo)
    for n, c in enumerate(cases):
        post, plen = O.pseudo_posterior(c["post_ids"], geo.ctc_vocab)
        emb, mask, _, _ = O.merge(O.projector(sd, post, mode), plen, sd["llm.model.embed_tokens.weight"][c["ids"]], c["ids"],
                                  c["am"], None, geo.speech_id)
        toks = O.beam_search_generate(sd, emb.detach(), mask, gd, mode=mode, **c["kw"])
        assert np.array_equal(toks.numpy(), c["tokens"]), (n, toks, c["tokens"])
        if n < 4:                                               # ... and the KV-cache form on the first cases
            toks = O.beam_search_generate(sd, emb.detach(), mask, gd, mode=mode, kv_cache=True, **c["kw"])
            assert np.array_equal(toks.numpy(), c["tokens"]), (n, toks, c["tokens"])


def test_generate_unfiltered_cases_vs_reference_fp32():
    """tests/golden/mid_generate_fp32.npz: 24 random decode cases kept WITHOUT any stability selection, tokens of the REAL
    reference's fp32 ``generate``.  The fp32-mode oracle reproduces every one (whole-sequence form; the KV-cache form on every
    third case); the bf16-mode oracle only the ones the generator recorded -- the set does contain rounding-sensitive cases.
    Case 5's first utterance pins HF's ``min_length`` semantics under ``inputs_embeds`` (EOS banned for max(min_length - S, 0)
    positions only: the reference emits EOS at once although min_length = 6)."""
    import dataclasses

    from conftest import decode_fp32_cases
    geo, sd, cases, bf16_agrees = decode_fp32_cases()
    gd = dataclasses.asdict(geo)
    assert len(cases) == 24 and 0 < sum(bf16_agrees) < len(cases)
    assert cases[5]["kw"]["min_length"] == 6 and (cases[5]["tokens"][0] == geo.eos_id).all()
    for n, c in enumerate(cases):
        post, plen = O.pseudo_posterior(c["post_ids"], geo.ctc_vocab)
        for mode in ("fp32", "bf16"):
            emb, mask, _, _ = O.merge(O.projector(sd, post, mode), plen, sd["llm.model.embed_tokens.weight"][c["ids"]], c["ids"], c["am"],
                                      None, geo.speech_id)
            toks = O.beam_search_generate(sd, emb.detach(), mask, gd, mode=mode, **c["kw"])
            same = toks.shape == c["tokens"].shape and np.array_equal(toks.numpy(), c["tokens"])
            assert same == (True if mode == "fp32" else bf16_agrees[n]), (n, mode, toks, c["tokens"])
            if mode == "fp32" and n % 3 == 0:
                toks = O.beam_search_generate(sd, emb.detach(), mask, gd, mode=mode, kv_cache=True, **c["kw"])
                assert np.array_equal(toks.numpy(), c["tokens"]), (n, "kv_cache", toks, c["tokens"])


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_generate_lora_margin_cases_vs_reference(mode):
    """The 7 rounding-stable decode cases of the LoRA-ADAPTED model (oracle/make_golden_generate_lora_margin.py): the oracle's
    beam search on the weights with W + s B A merged in reproduces the tokens of the REAL reference's generate() with the LoRA
    formula applied by hand to its decoder (Multitask/ps-slm.py:199-216, 640-673) -- in fp32 on the fp32 merge and in bf16 mode on
    the product's arithmetic (bf16 base weight + fp32 B A, rounded once more)."""
    import dataclasses

    from conftest import decode_lora_margin_cases
    from ps_slm_amd.lora import key_of
    geo, cfg, sd, lsd, cases = decode_lora_margin_cases()
    gd = dataclasses.asdict(geo)
    sdm = dict(sd)
    for l in range(geo.llm_layers):
        for t in cfg.target_modules:
            k = f"llm.model.layers.{l}.{'mlp' if t in ('gate_proj', 'up_proj', 'down_proj') else 'self_attn'}.{t}.weight"
            w = sd[k].double() if mode == "fp32" else sd[k].bfloat16().double()
            sdm[k] = (w + cfg.scaling * (lsd[key_of(l, t, "B")].double() @ lsd[key_of(l, t, "A")].double())).float()
    for n, c in enumerate(cases):
        post, plen = O.pseudo_posterior(c["post_ids"], geo.ctc_vocab)
        emb, mask, _, _ = O.merge(O.projector(sd, post, mode), plen, sd["llm.model.embed_tokens.weight"][c["ids"]], c["ids"],
                                  c["am"], None, geo.speech_id)
        toks = O.beam_search_generate(sdm, emb.detach(), mask, gd, mode=mode, **c["kw"])
        assert np.array_equal(toks.numpy(), c["tokens"]), (n, toks, c["tokens"])


@pytest.mark.parametrize("k", [1, 2])
def test_linear_projector_vs_reference(k):
    """encoder_projector="linear" (EncoderProjectorConcat, projector.py:28-49), ds_rate 1 and 2, through the REAL reference at
    the mid geometry: the oracle's restatement reproduces loss, accuracy, logits and projector gradients."""
    import dataclasses

    from conftest import linear_projector_case
    geo, sd, batch, z = linear_projector_case(k)
    out, grads = O.loss_and_projector_grads(sd, batch, dataclasses.asdict(geo), "fp32")
    close(out["loss"], z["loss"])
    close(out["acc"], z["acc"])
    cols = torch.from_numpy(z["cols"])
    close(out["logits"][:, :, cols], z["logits_cols"], rtol=2e-4, atol=2e-5)
    close(grads["encoder_projector.linear1.bias"], z["grad.linear1.bias"], rtol=2e-4, atol=1e-7)
    close(grads["encoder_projector.linear2.bias"], z["grad.linear2.bias"], rtol=2e-4, atol=1e-7)
    close(grads["encoder_projector.linear2.weight"][::16], z["grad.linear2.weight.rows16"], rtol=2e-4, atol=1e-7)
    close(grads["encoder_projector.linear1.weight"][::64], z["grad.linear1.weight.rows64"], rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize("k", [1, 2])
def test_cov1d_projector_vs_reference(k):
    """encoder_projector="cov1d-linear" (EncoderProjectorCov1d, projector.py:53-73), kernel = stride = 1 and 2, through the REAL
    reference at the mid geometry: the oracle's restatement reproduces loss, accuracy, logits and projector gradients."""
    import dataclasses

    from conftest import cov1d_projector_case
    geo, sd, batch, z = cov1d_projector_case(k)
    out, grads = O.loss_and_projector_grads(sd, batch, dataclasses.asdict(geo), "fp32")
    close(out["loss"], z["loss"])
    close(out["acc"], z["acc"])
    cols = torch.from_numpy(z["cols"])
    close(out["logits"][:, :, cols], z["logits_cols"], rtol=2e-4, atol=2e-5)
    close(grads["encoder_projector.conv1d.bias"], z["grad.conv1d.bias"], rtol=2e-4, atol=1e-7)
    close(grads["encoder_projector.conv1d.weight"][::8], z["grad.conv1d.weight.rows8"], rtol=2e-4, atol=1e-7)
    close(grads["encoder_projector.linear1.bias"], z["grad.linear1.bias"], rtol=2e-4, atol=1e-7)
    close(grads["encoder_projector.linear2.bias"], z["grad.linear2.bias"], rtol=2e-4, atol=1e-7)
    close(grads["encoder_projector.linear2.weight"][::16], z["grad.linear2.weight.rows16"], rtol=2e-4, atol=1e-7)
    close(grads["encoder_projector.linear1.weight"][::64], z["grad.linear1.weight.rows64"], rtol=2e-4, atol=1e-7)


def test_cross_attention_projector_vs_reference():
    """encoder_projector="cross-attention" (EncoderProjectorCTCCA, projector.py:104-126: 8 heads over the LLM's embedding table)
    through the REAL reference at llm_dim 512: the oracle's restatement reproduces loss, accuracy, logits and the W_q gradient."""
    import dataclasses

    from conftest import ca_projector_case
    geo, sd, batch, z = ca_projector_case()
    out, grads = O.loss_and_projector_grads(sd, batch, dataclasses.asdict(geo), "fp32")
    close(out["loss"], z["loss"])
    close(out["acc"], z["acc"])
    cols = torch.from_numpy(z["cols"])
    close(out["logits"][:, :, cols], z["logits_cols"], rtol=2e-4, atol=2e-5)
    close(grads["encoder_projector.W_q.weight"], z["grad.W_q.weight"], rtol=5e-4, atol=1e-8)


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/Multitask"), reason="needs the reference tree (build container only)")
def test_reference_cannot_run_its_q_former():
    """FINDING behind the plugin's NotImplementedError for encoder_projector='q-former': the REAL reference (imported) builds
    EncoderProjectorQFormer but its slam_model_asr.forward cannot call it -- the projector's forward needs `atts`
    (Multitask/model/projector.py:91), every call site passes one argument and reads `.k` (ps-slm.py:482); with cross_attn the
    embedding table lands in `atts` and the merge fails.  Nothing to pin an implementation against."""
    import contextlib
    import dataclasses
    import io

    from oracle.ref_import import Cfg, build_reference_model, load_reference
    from ps_slm_amd.model import Geometry
    from ps_slm_amd.synthetic import MID_GEOMETRY, synthetic_text_batch
    _, _, proj = load_reference()
    geo = Geometry.from_dict(MID_GEOMETRY)
    model = build_reference_model(dataclasses.asdict(geo), 0, dict(gt_emb=True, gt_emb_noise=False))
    model.encoder_projector = proj.EncoderProjectorQFormer(Cfg(encoder_projector="q-former", encoder_dim=geo.ctc_vocab, llm_dim=geo.llm_dim,
                                                               qformer_layers=1, query_len=8))
    assert not hasattr(model.encoder_projector, "k")
    batch = synthetic_text_batch(geo, 2, seed=1, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=12, noise=False)
    call = dict(input_ids=batch["input_ids"], input_features=batch["input_features"], attention_mask=batch["attention_mask"],
                input_feature_length=batch["input_feature_length"], GT=[" ".join(map(str, p)) for p in batch["post_ids"]],
                labels=batch["labels"])
    with contextlib.redirect_stdout(io.StringIO()):
        model.cross_attn = False
        with pytest.raises(TypeError, match="atts"):
            model(**call)
        model.cross_attn = True
        with pytest.raises((RuntimeError, ValueError, TypeError, AttributeError)):
            model(**call)
    from fake_ops import FakeOps
    from ps_slm_amd.config import ModelConfig, TrainConfig
    from ps_slm_amd.ps_slm import model_factory
    with pytest.raises(NotImplementedError, match="q-former"):
        model_factory(TrainConfig(freeze_llm=True, gt_emb=True, ctc_posterior=True), ModelConfig(llm_path="synthetic:mid", encoder_projector="q-former", llm_dim=256),
                      device="cpu", ops=FakeOps())
