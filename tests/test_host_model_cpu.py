"""Drives the REAL host code (ps_slm_amd.model / merge) on CPU through the FakeOps test double and checks it
against the oracle.  This pins buffer bookkeeping, the forward/backward schedule and the merge plan; the HIP
kernels themselves are checked against the same double in tests/test_gpu_*.py."""
import dataclasses

import numpy as np
import pytest
import torch

from fake_ops import FakeOps
from oracle import tasu_oracle as O
from ps_slm_amd.model import Geometry, TasuModel
from ps_slm_amd.synthetic import MID_GEOMETRY, random_state_dict, synthetic_text_batch


def build(geo, sd):
    m = TasuModel(geo, FakeOps(), "cpu")
    m.load_reference_state_dict(sd)
    return m


def run_text(model, batch):
    st = model.prepare_text(batch["input_ids"], batch["attention_mask"], batch["labels"], batch["post_ids"],
                            batch.get("alphas"), batch.get("keeps"))
    model.forward_projector_text(st)
    model.forward_llm(st)
    model.backward(st)
    return st


@pytest.fixture(scope="module")
def mid():
    geo = Geometry.from_dict(MID_GEOMETRY)
    sd = random_state_dict(geo, 7, with_encoder=False)
    return geo, sd


@pytest.mark.parametrize("ragged,noise,drop", [(False, False, 0.0), (True, True, 0.0), (True, True, 0.2)])
def test_text_step_matches_oracle_bf16(mid, ragged, noise, drop):
    geo, sd = mid
    batch = synthetic_text_batch(geo, 3, seed=11, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=8,
                                 noise=noise, drop_prob=drop, ragged=ragged)
    model = build(geo, sd)
    st = run_text(model, batch)
    gd = dataclasses.asdict(geo)
    out, grads = O.loss_and_projector_grads(sd, batch, gd, "bf16")
    res = st.dev["loss_out"]
    assert abs(float(res[0]) - float(out["loss"])) < 5e-3
    assert abs(float(res[1]) - float(out["acc"])) < 1e-6 + 1.0 / max(st.plan.count, 1)
    lg = model.logits_view(st).float()
    ref = out["logits"].detach()
    valid = out["mask"]
    assert np.array_equal(st.plan.key_mask[:, : st.S].astype(bool), valid.numpy())
    err = (lg - ref)[valid].abs().max() / ref[valid].abs().max()
    assert err < 2e-2, err
    mine = model.projector_grads()
    for k, g in grads.items():
        cos = torch.nn.functional.cosine_similarity(mine[k].flatten(), g.flatten(), dim=0)
        rel = (mine[k] - g).norm() / g.norm()
        assert cos > 0.999 and rel < 5e-2, (k, float(cos), float(rel))


def test_text_step_close_to_fp32_oracle(mid):
    geo, sd = mid
    batch = synthetic_text_batch(geo, 2, seed=5, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=8,
                                 ragged=True)
    model = build(geo, sd)
    st = run_text(model, batch)
    out, grads = O.loss_and_projector_grads(sd, batch, dataclasses.asdict(geo), "fp32")
    assert abs(float(st.dev["loss_out"][0]) - float(out["loss"])) < 2e-2
    mine = model.projector_grads()
    for k, g in grads.items():
        cos = torch.nn.functional.cosine_similarity(mine[k].flatten(), g.flatten(), dim=0)
        assert cos > 0.995, (k, float(cos))


def test_left_padding_batch(mid):
    geo, sd = mid
    batch = synthetic_text_batch(geo, 2, seed=3, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=8,
                                 noise=False, ragged=True)
    # turn right padding into left padding
    ids, am, lab = batch["input_ids"].clone(), batch["attention_mask"].clone(), batch["labels"].clone()
    for b in range(ids.shape[0]):
        n = int(am[b].sum())
        L = ids.shape[1]
        ids[b] = torch.cat([ids[b, n:], ids[b, :n]])
        am[b] = torch.cat([am[b, n:], am[b, :n]])
        lab[b] = torch.cat([lab[b, n:], lab[b, :n]])
    batch.update(input_ids=ids, attention_mask=am, labels=lab)
    model = build(geo, sd)
    st = run_text(model, batch)
    assert st.plan.left_padding
    out, grads = O.loss_and_projector_grads(sd, batch, dataclasses.asdict(geo), "bf16")
    assert abs(float(st.dev["loss_out"][0]) - float(out["loss"])) < 5e-3
    mine = model.projector_grads()
    for k, g in grads.items():
        assert torch.nn.functional.cosine_similarity(mine[k].flatten(), g.flatten(), dim=0) > 0.999, k


def test_state_dict_roundtrip(mid):
    geo, sd = mid
    model = build(geo, sd)
    out = model.projector_state_dict()
    for k, v in out.items():
        assert v.shape == sd[k].shape
        torch.testing.assert_close(v, sd[k])
    assert model.proj.num_parameters() == sum(sd[k].numel() for k in out)


# ------------------------------------------------------------------ audio path (encoder -> CTC -> PSD -> projector -> LLM)
def test_audio_step_matches_oracle_and_reference_golden():
    from conftest import load_npz
    z = load_npz("mid_audio")
    geo = Geometry.from_dict(MID_GEOMETRY)
    sd = random_state_dict(geo, int(z["seed_w"]), with_encoder=True)
    batch = synthetic_text_batch(geo, 3, seed=int(z["seed_b"]), prompt_len=9, n_audio=21, target_len=17, speech_pos=4,
                                 feat_frames=12, noise=True, drop_prob=0.15, ragged=True)
    model = build(geo, sd)
    st = model.prepare_audio(batch["input_ids"], batch["attention_mask"], batch["labels"], batch["input_features"],
                             batch["input_feature_length"])
    model.forward_llm(st)
    model.backward(st)
    gd = dataclasses.asdict(geo)
    out, grads = O.loss_and_projector_grads(sd, batch, gd, "bf16", audio=True)
    assert abs(float(st.dev["loss_out"][0]) - float(out["loss"].detach())) < 1e-2
    # PSD plan: same kept-frame counts as the oracle (and therefore as the reference, which pins the oracle)
    post, _, lens = O.audio_front(sd, batch["input_features"], batch["input_feature_length"], geo.enc_heads, geo.enc_kernel, "bf16")
    _, pl = O.psd(post, lens, post, 0)
    assert np.array_equal(st.dev["psd_lens"], pl.numpy())
    mine = model.projector_grads()
    for k, g in grads.items():
        cos = torch.nn.functional.cosine_similarity(mine[k].flatten(), g.flatten(), dim=0)
        assert cos > 0.995, (k, float(cos))
    # real-reference golden (fp32) within the stated bf16 tolerance
    assert abs(float(st.dev["loss_out"][0]) - float(z["loss"])) < 3e-2
    for k, g in mine.items():
        short = "grad." + k[len("encoder_projector."):]
        if short in z:
            assert torch.nn.functional.cosine_similarity(g.flatten(), torch.from_numpy(z[short]).flatten(), dim=0) > 0.99, k


def test_audio_psd_step_vs_reference_golden():
    from conftest import mid_audio_psd_case
    geo, sd, batch, z = mid_audio_psd_case()
    model = build(geo, sd)
    st = model.prepare_audio(batch["input_ids"], batch["attention_mask"], batch["labels"], batch["input_features"],
                             batch["input_feature_length"])
    model.forward_llm(st)
    model.backward(st)
    assert np.array_equal(st.dev["psd_lens"], z["psd_lens"])
    assert abs(float(st.dev["loss_out"][0]) - float(z["loss"])) < 3e-2
    mine = model.projector_grads()
    for k, g in mine.items():
        short = "grad." + k[len("encoder_projector."):]
        if short in z:
            assert torch.nn.functional.cosine_similarity(g.flatten(), torch.from_numpy(z[short]).flatten(), dim=0) > 0.99, k


# ------------------------------------------------------------------ decode: beam-4 generate
def gen_case():
    from conftest import load_npz, mid_audio_psd_case, split_flat
    geo, sd, batch, _ = mid_audio_psd_case()
    z = load_npz("mid_generate_beam4")
    return geo, sd, batch, z, split_flat(z["post_ids_flat"], z["post_lens"])


def test_generate_beam4_text_and_audio_vs_reference_tokens():
    """Product decode loop (KV cache, per-row top-k + host beam bookkeeping) through the CPU double vs the token
    ids the REAL reference's generate() produced (fp32).  bf16 arithmetic may legitimately flip a near-tie, so the
    bf16 oracle is the exact comparator and the fp32 reference must agree on a long common prefix."""
    from ps_slm_amd.decode import beam_search_generate
    geo, sd, batch, z, word_ids = gen_case()
    gd = dataclasses.asdict(geo)
    ids, am = torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"])
    model = build(geo, sd)
    # text path
    st = model.prepare_text(ids, am, None, word_ids, None, None)
    model.forward_projector_text(st)
    toks = beam_search_generate(model, st, max_new_tokens=16).numpy()
    post, plen = O.pseudo_posterior(word_ids, geo.ctc_vocab)
    emb, mask, _, _ = O.merge(O.projector(sd, post, "bf16"), plen, sd["llm.model.embed_tokens.weight"][ids], ids, am, None, geo.speech_id)
    ref16 = O.beam_search_generate(sd, emb.detach(), mask, gd, max_new_tokens=16, mode="bf16").numpy()
    assert np.array_equal(toks, ref16), (toks, ref16)
    common = (toks == z["tokens_text"]).cumprod(1).sum(1)
    assert (common >= 8).all(), (toks, z["tokens_text"])
    # audio path
    feats, fl = batch["input_features"][:2], batch["input_feature_length"][:2]
    st = model.prepare_audio(ids, am, None, feats, fl)
    toks = beam_search_generate(model, st, max_new_tokens=16).numpy()
    common = (toks == z["tokens_audio"]).cumprod(1).sum(1)
    assert (common >= 8).all(), (toks, z["tokens_audio"])


@pytest.mark.parametrize("lpw,min_len", [(1.0, 1), (2.0, 1), (0.5, 4), (1.0, 6)])
@pytest.mark.parametrize("nb", [1, 2, 3, 4])
def test_beam_state_matches_oracle_on_random_scores(nb, lpw, min_len):
    """Host beam bookkeeping vs the oracle's tensor formulation (the loop of oracle/tasu_oracle.py::beam_search_generate, which
    tests/test_oracle_golden.py pins against the reference's generate() for these beam counts, length penalties and minimum
    lengths) on a synthetic score stream with EOS events."""
    from ps_slm_amd.decode import BeamState
    B, V, T, eos = 3, 50, 12, 7
    g = torch.Generator().manual_seed(0)
    table = torch.randn(64, V, generator=g) * 2.0
    table[:, eos] += 1.5                                      # make EOS competitive so beams really finish

    class FakeW(dict):
        pass
    # drive both implementations with logits that depend only on (step, last token)
    def logits_for(tokens, t):
        last = tokens[:, t - 1] if t > 0 else torch.zeros(tokens.shape[0], dtype=torch.long)
        return table[(last * 7 + t) % 64]
    state = BeamState(B, nb, T, eos, eos, lpw, min_len)
    while not state.done:
        t = state.cur
        seqs = torch.from_numpy(state.run_seq).view(B * nb, -1)
        lp = torch.log_softmax(logits_for(seqs, t), -1)
        if state.ban_eos():
            lp[:, eos] = float("-inf")
        v, i = torch.sort(lp, dim=-1, descending=True, stable=True)
        state.update(v[:, : 2 * nb].numpy().reshape(B, nb, -1), i[:, : 2 * nb].numpy().reshape(B, nb, -1))
    mine = state.result()
    # oracle formulation (full [nb*V] top-k)
    NEG = -1.0e9
    run_seq = torch.full((B, nb, T), eos, dtype=torch.long)
    fin_seq, run_sc = run_seq.clone(), torch.zeros(B, nb)
    run_sc[:, 1:] = NEG
    fin_sc, fin_len = torch.full((B, nb), NEG), torch.zeros(B, nb, dtype=torch.long)
    is_fin, unsat = torch.zeros(B, nb, dtype=torch.bool), torch.ones(B, 1, dtype=torch.bool)
    cur = 0
    while True:
        lp = torch.log_softmax(logits_for(run_seq.view(B * nb, -1), cur), -1)
        if cur < min_len:
            lp[:, eos] = float("-inf")
        acc = (lp.view(B, nb, V) + run_sc[:, :, None]).view(B, nb * V)
        top_lp, top_ix = torch.topk(acc, 2 * nb)
        beam, tok = top_ix // V, top_ix % V
        cand = torch.gather(run_seq, 1, beam[:, :, None].expand(-1, -1, T)).clone()
        cand[:, :, cur] = tok
        stop = (tok == eos) | (cur + 1 >= T)
        run_lp = top_lp + stop.float() * NEG
        nxt = torch.topk(run_lp, nb)[1]
        run_seq = torch.gather(cand, 1, nxt[:, :, None].expand(-1, -1, T))
        run_sc = torch.gather(run_lp, 1, nxt)
        just = stop & (torch.arange(2 * nb) < nb)[None]
        sc = top_lp / ((cur + 1) ** lpw) + (~unsat).float() * NEG + (~just).float() * NEG
        keep = torch.topk(torch.cat([fin_sc, sc], 1), nb)[1]
        fin_seq = torch.gather(torch.cat([fin_seq, cand], 1), 1, keep[:, :, None].expand(-1, -1, T))
        fin_len = torch.gather(torch.cat([fin_len, torch.full((B, 2 * nb), cur + 1)], 1), 1, keep)
        is_fin = torch.gather(torch.cat([is_fin, just], 1), 1, keep)
        fin_sc = torch.gather(torch.cat([fin_sc, sc], 1), 1, keep)
        cur += 1
        worst = torch.where(is_fin, fin_sc.min(1, keepdim=True)[0], torch.full_like(fin_sc, NEG))
        unsat = unsat & (run_sc[:, :1] / (cur ** lpw) > worst).any(-1, keepdim=True)
        if not (bool(unsat.any()) and not bool(stop.all())):
            break
    ref = fin_seq[:, 0, : int(fin_len[:, 0].max())].numpy()
    assert np.array_equal(mine, ref), (mine, ref)
    assert (mine == eos).any(), "the stream must exercise the EOS / finished-beam path"


def test_synthetic_inference_transcripts_survive_reference_cleaning():
    """The synthetic test split must give generate() transcripts that are non-empty after the reference's regex
    (ps-slm.py:592-596) -- digit strings are wiped by it."""
    import re
    from ps_slm_amd.finetune_deepspeed import SyntheticDataset
    from ps_slm_amd.ps_slm import SyntheticSentencePiece
    geo = Geometry.from_dict(MID_GEOMETRY)
    ds = SyntheticDataset(geo, 2, 1, 0, inference=True)
    b = ds.collator(next(iter(ds)))
    sp = SyntheticSentencePiece(geo.ctc_vocab)
    for t in b["targets"]:
        cleaned = re.sub(r"[^A-Za-z\s.,!?]+", "", t).lower().strip()
        assert cleaned == t and len(sp.encode(cleaned)) == len(t.split()) > 0


def test_model_factory_from_hf_directory(tmp_path):
    """llm_path pointing at a HF-style directory (config.json + *.safetensors, no tokenizer files): geometry from the
    config, weights through load_hf_llm_state_dict, projector checkpoint through ckpt_path -- same loss as the model built
    directly from the same tensors."""
    import json
    from safetensors.torch import save_file
    from ps_slm_amd.config import ModelConfig, TrainConfig
    from ps_slm_amd.ps_slm import model_factory
    # the projector bottleneck is not an HF config field: the reference hard-codes 2048 (projector.py:141), the default
    geo = Geometry.from_dict(dict(MID_GEOMETRY, bottleneck=Geometry().bottleneck))
    sd = random_state_dict(geo, 77, with_encoder=False)
    hf = tmp_path / "qwen_mid"
    hf.mkdir()
    json.dump(dict(vocab_size=geo.llm_vocab, hidden_size=geo.llm_dim, intermediate_size=geo.llm_inter,
                   num_hidden_layers=geo.llm_layers, num_attention_heads=geo.llm_heads, num_key_value_heads=geo.llm_kv_heads,
                   head_dim=128, rope_theta=geo.rope_theta, rms_norm_eps=geo.rms_eps, tie_word_embeddings=True),
              open(hf / "config.json", "w"))
    save_file({k[4:]: v.contiguous() for k, v in sd.items() if k.startswith("llm.")}, str(hf / "model.safetensors"))
    # a real tokenizer directory: 900 words + <eos>; model_factory adds <speech> (-> id 901) like ps-slm.py:133-140 and
    # takes the special ids from the tokenizer
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {f"w{i}": i for i in range(900)}
    vocab["<eos>"] = 900
    t = Tokenizer(models.WordLevel(vocab, unk_token="w0"))
    t.pre_tokenizer = pre_tokenizers.Whitespace()
    PreTrainedTokenizerFast(tokenizer_object=t, eos_token="<eos>").save_pretrained(str(hf))
    ckpt = tmp_path / "projector.pt"
    torch.save({k: v for k, v in sd.items() if k.startswith("encoder_projector.")}, ckpt)
    tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, gt_emb_noise=False, ctc_posterior=True, do_psd=True)
    mc = ModelConfig(llm_path=str(hf), llm_dim=geo.llm_dim, encoder_projector="linear-silu", encoder_dim=geo.ctc_vocab)
    model, tok = model_factory(tc, mc, ops=FakeOps(), device="cpu", ckpt_path=str(ckpt))
    assert tok.default_speech_token == 901 and tok.pad_token_id == tok.eos_token_id == 900
    assert (model.core.geo.speech_id, model.core.geo.eos_id) == (901, 900)
    geo.speech_id, geo.eos_id = 901, 900
    ref = TasuModel(geo, FakeOps(), "cpu")
    ref.load_reference_state_dict(sd)
    batch = synthetic_text_batch(geo, 2, seed=5, prompt_len=9, n_audio=13, target_len=11, speech_pos=4, feat_frames=12, noise=False)
    def loss_of(core):
        st = core.prepare_text(batch["input_ids"], batch["attention_mask"], batch["labels"], batch["post_ids"], None, None)
        core.forward_projector_text(st)
        core.forward_llm(st, need_backward=False)
        return float(st.dev["loss_out"][0])
    assert model.core.geo.llm_layers == geo.llm_layers and model.core.geo.llm_vocab == geo.llm_vocab
    assert abs(loss_of(model.core) - loss_of(ref)) < 1e-6


def test_model_factory_audio_branch_from_checkpoints(tmp_path):
    """Audio recipe through model_factory: HF LLM directory + funasr-style encoder directory (config.yaml + model.pt).
    Same loss as the model loaded directly from the reference-named state dict."""
    import json
    import yaml
    from safetensors.torch import save_file
    from ps_slm_amd.config import ModelConfig, TrainConfig
    from ps_slm_amd.ps_slm import model_factory
    geo = Geometry.from_dict(dict(MID_GEOMETRY, bottleneck=Geometry().bottleneck))
    sd = random_state_dict(geo, 78, with_encoder=True)
    hf, enc = tmp_path / "llm", tmp_path / "sensevoice"
    hf.mkdir()
    enc.mkdir()
    json.dump(dict(vocab_size=geo.llm_vocab, hidden_size=geo.llm_dim, intermediate_size=geo.llm_inter,
                   num_hidden_layers=geo.llm_layers, num_attention_heads=geo.llm_heads, num_key_value_heads=geo.llm_kv_heads,
                   head_dim=128, rope_theta=geo.rope_theta, rms_norm_eps=geo.rms_eps, tie_word_embeddings=True),
              open(hf / "config.json", "w"))
    save_file({k[4:]: v.contiguous() for k, v in sd.items() if k.startswith("llm.")}, str(hf / "model.safetensors"))
    yaml.safe_dump(dict(input_size=geo.feat_dim,
                        encoder_conf=dict(output_size=geo.enc_dim, attention_heads=geo.enc_heads, linear_units=geo.enc_ffn,
                                          num_blocks=geo.enc_blocks, tp_blocks=geo.enc_tp_blocks, kernel_size=geo.enc_kernel,
                                          sanm_shfit=0)), open(enc / "config.yaml", "w"))
    torch.save({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}, enc / "model.pt")
    ckpt = tmp_path / "projector.pt"
    torch.save({k: v for k, v in sd.items() if k.startswith("encoder_projector.")}, ckpt)
    tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=False, ctc_posterior=True, do_psd=True)
    mc = ModelConfig(llm_path=str(hf), llm_dim=geo.llm_dim, encoder_path=str(enc), encoder_projector="linear-silu",
                     encoder_dim=geo.ctc_vocab)
    model, _ = model_factory(tc, mc, ops=FakeOps(), device="cpu", ckpt_path=str(ckpt))
    model.core.geo.speech_id, model.core.geo.eos_id = geo.speech_id, geo.eos_id
    assert dataclasses.asdict(model.core.geo) == dataclasses.asdict(geo)
    ref = TasuModel(geo, FakeOps(), "cpu")
    ref.load_reference_state_dict(sd)
    batch = synthetic_text_batch(geo, 2, seed=6, prompt_len=9, n_audio=13, target_len=11, speech_pos=4, feat_frames=24, noise=False)

    def loss_of(core):
        st = core.prepare_audio(batch["input_ids"], batch["attention_mask"], batch["labels"], batch["input_features"],
                                batch["input_feature_length"], do_psd=True)
        core.forward_llm(st, need_backward=False)
        return float(st.dev["loss_out"][0])
    assert abs(loss_of(model.core) - loss_of(ref)) < 1e-6


def test_encoder_tokenizer_from_sentencepiece_model(tmp_path):
    """setup_encoder_tokenizer picks up funasr's BPE model file (Multitask/model/tokenizer.py) when it is there."""
    import sentencepiece as spm
    from ps_slm_amd.config import ModelConfig
    from ps_slm_amd.ps_slm import setup_encoder_tokenizer
    corpus = tmp_path / "corpus.txt"
    corpus.write_text("\n".join(f"hello world this is sentence number {i} of the tiny corpus" for i in range(200)))
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(tmp_path / "chn_jpn_yue_eng_ko_spectok.bpe"),
                                   vocab_size=60, model_type="bpe", minloglevel=2)
    geo = Geometry.from_dict(MID_GEOMETRY)
    tok = setup_encoder_tokenizer(ModelConfig(encoder_path=str(tmp_path)), geo)
    ids = tok.encode("hello tiny world")
    assert tok.vocab_size == 60 and len(ids) > 0 and all(isinstance(i, int) and 0 <= i < 60 for i in ids)
    assert setup_encoder_tokenizer(ModelConfig(encoder_path=str(tmp_path / "nope")), geo).vocab_size == geo.ctc_vocab


@pytest.mark.parametrize("nb", [1, 2, 3])
def test_generate_other_beam_counts_match_bf16_oracle(nb):
    """num_beams 1 (greedy), 2 and 3 (top-k widths 2, 4, 6): product decode loop through the CPU double against the bf16
    oracle.  The bookkeeping itself is compared exactly in test_beam_state_matches_oracle_on_random_scores; here the two
    bf16 emulations (KV-cache loop vs full re-run) may part at a near-tie of this random-init model, so a common prefix
    of 3 tokens is required for every utterance and exact equality for most."""
    from ps_slm_amd.decode import beam_search_generate
    geo, sd, batch, z, word_ids = gen_case()
    gd = dataclasses.asdict(geo)
    ids, am = torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"])
    model = build(geo, sd)
    st = model.prepare_text(ids, am, None, word_ids, None, None)
    model.forward_projector_text(st)
    toks = beam_search_generate(model, st, num_beams=nb, max_new_tokens=12).numpy()
    post, plen = O.pseudo_posterior(word_ids, geo.ctc_vocab)
    emb, mask, _, _ = O.merge(O.projector(sd, post, "bf16"), plen, sd["llm.model.embed_tokens.weight"][ids], ids, am, None, geo.speech_id)
    ref16 = O.beam_search_generate(sd, emb.detach(), mask, gd, num_beams=nb, max_new_tokens=12, mode="bf16").numpy()
    n = min(toks.shape[1], ref16.shape[1])
    common = (toks[:, :n] == ref16[:, :n]).cumprod(1).sum(1)
    assert (common >= 3).all() and (common == n).any(), (toks, ref16)


def test_cps_noise_draws_and_posterior_vs_reference():
    """20 seeded calls of the REAL reference's ctc_pseudo_posterior_noise (oracle/make_golden_noise.py; the last 8 with
    insert_prob > 0): from the same torch seed the plugin's draw_noise / draw_noise_rows must make the same draws in the same
    order, and the posterior built from them (oracle arithmetic and the product's posterior kernel through the CPU double) must
    equal the reference's."""
    from conftest import load_npz, split_flat
    from ps_slm_amd.config import ModelConfig, TrainConfig
    from ps_slm_amd.ps_slm import model_factory
    z = load_npz("cps_noise_random")
    tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, gt_emb_noise=True, ctc_posterior=True, do_psd=True)
    mc = ModelConfig(llm_path="synthetic:mid", encoder_projector="linear-silu", llm_dim=256)
    model, _ = model_factory(tc, mc, device="cpu", ops=FakeOps(), init_seed=3)
    core, V = model.core, int(load_npz("geometry")["ctc_vocab"])
    n_insert_cases = 0
    for n in range(int(z["n_cases"])):
        ids = [list(map(int, p)) for p in split_flat(z[f"c{n}_ids_flat"], z[f"c{n}_ids_lens"])]
        model.drop_prob, model.smooth_low, model.smooth_high = (float(v) for v in z[f"c{n}_params"])
        model.insert_prob = float(z[f"c{n}_insert_prob"])
        torch.manual_seed(int(z[f"c{n}_seed"]))
        ref = torch.from_numpy(z[f"c{n}_posterior"])
        if model.insert_prob == 0.0:
            alphas, keeps = model.draw_noise(ids)
            post, lens = O.pseudo_posterior(ids, V, alphas, keeps)
            row_ids = [[i for i, k in zip(u, kk) if k] for u, kk in zip(ids, keeps)]
            row_alphas = [[a] * len(r) for a, r in zip(alphas, row_ids)]
        else:
            n_insert_cases += 1
            row_ids, row_alphas = model.draw_noise_rows(ids, blank_id=0)
            lens = torch.tensor([len(r) for r in row_ids])
            post = torch.zeros(len(ids), int(lens.max()), V)
            for b, (r, al) in enumerate(zip(row_ids, row_alphas)):
                for i, (tok_id, a) in enumerate(zip(r, al)):
                    post[b, i] = a / V
                    post[b, i, tok_id] += 1 - a
        assert np.array_equal(lens.numpy(), z[f"c{n}_lens"]), n
        torch.testing.assert_close(post, ref, rtol=1e-6, atol=1e-8)
        # the product's posterior rows (kernel semantics through the CPU double) at the mid geometry's vocabulary: same ids
        # (all < 203), same draws -> the same rows up to the 1 / V smoothing floor, checked through the argmax and the peak
        B, Lmax = len(ids), int(lens.max())
        tok = torch.full((B, 3), 5, dtype=torch.long)
        tok[:, 1] = core.geo.speech_id
        st = core.prepare_text(tok, torch.ones(B, 3, dtype=torch.bool), None, row_ids, None, None, row_alphas=row_alphas)
        core.forward_projector_text(st)
        rows = st.dev["post"][: B * Lmax].view(B, Lmax, -1)[:, :, : core.geo.ctc_vocab].float()
        live = torch.arange(Lmax)[None, :] < lens[:, None]
        assert torch.equal(rows.argmax(-1)[live], ref.argmax(-1)[live]), n
        a = torch.zeros(B, Lmax)
        for b, al in enumerate(row_alphas):
            a[b, : len(al)] = torch.tensor(al)
        a = a[live]
        torch.testing.assert_close(rows.max(-1).values[live], (1 - a) + a / core.geo.ctc_vocab, rtol=1e-6, atol=1e-7)
        assert float(rows[~live].abs().max()) == 0.0 if (~live).any() else True
    assert n_insert_cases == 8
    # the forward of the plugin takes the insertion path when insert_prob is set
    model.insert_prob, model.drop_prob = 0.5, 0.1
    raw = synthetic_text_batch(core.geo, 2, seed=3, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=8, noise=False)
    torch.manual_seed(5)
    out, _ = model(input_ids=raw["input_ids"], attention_mask=raw["attention_mask"], labels=raw["labels"],
                   GT=[" ".join(map(str, p)) for p in raw["post_ids"]])
    assert torch.isfinite(out.loss) and model.last_state.Ra > 2 * 21 * 0.9


def test_generate_margin_cases_exact_on_the_double():
    """Product decode loop (prefill, KV cache + row index, per-row top-k, beam bookkeeping) on the CPU double against the REAL
    reference's tokens on the 17 rounding-stable cases (3 of them 40-50 positions long) of tests/golden/mid_generate_margin.npz: exact equality."""
    from conftest import decode_margin_cases
    from ps_slm_amd.decode import beam_search_generate
    geo, sd, cases = decode_margin_cases()
    model = build(geo, sd)
    for n, c in enumerate(cases):
        st = model.prepare_text(c["ids"], c["am"], None, c["post_ids"], None, None)
        model.forward_projector_text(st)
        toks = beam_search_generate(model, st, eos_token_id=geo.eos_id, pad_token_id=geo.eos_id, **c["kw"]).numpy()
        assert np.array_equal(toks, c["tokens"]), (n, toks, c["tokens"])


@pytest.mark.parametrize("ragged", [False, True])
def test_labelled_rows_loss_head_equals_full_materialisation(mid, ragged):
    """keep_logits=False (the training step's throughput mode: final norm, lm_head, CE and lm_head dgrad over the labelled
    positions only) against keep_logits=True (logits for every position): same loss, accuracy and count (up to the order the row
    losses are summed in) and the same projector gradients up to fp32 summation order of the differently shaped GEMMs."""
    geo, sd = mid
    batch = synthetic_text_batch(geo, 3, seed=9, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=12,
                                 noise=True, drop_prob=0.1, ragged=ragged)
    full, lean = build(geo, sd), build(geo, sd)
    lean.keep_logits = False
    sf, sl = run_text(full, batch), run_text(lean, batch)
    assert sl.nL == int((sf.plan.shift_labels >= 0).sum()) and sl.nLp % 64 == 0 and lean.logits_view(sl) is None
    torch.testing.assert_close(sf.dev["loss_out"], sl.dev["loss_out"], rtol=1e-6, atol=0)
    assert float((full.proj.g - lean.proj.g).norm() / full.proj.g.norm()) < 5e-3     # bf16 roundings move with the summation order


@pytest.mark.parametrize("ragged", [False, True])
def test_tail_layer_on_labelled_rows_changes_nothing(mid, ragged):
    """TasuModel.tail_rows (the last decoder layer's MLP, the final norm and the loss head on the labelled rows only): every row
    is computed independently of the others after that layer's attention, so loss, accuracy, count AND the projector gradients
    equal the all-rows schedule's (on the double up to the summation order of torch's differently shaped CPU matmuls, which moves a
    few bf16 roundings; the HIP kernels, whose K order does not depend on the row count, are compared bit for bit in
    tests/test_gpu_model.py)."""
    geo, sd = mid
    batch = synthetic_text_batch(geo, 3, seed=11, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=12,
                                 noise=True, drop_prob=0.1, ragged=ragged)
    a, b = build(geo, sd), build(geo, sd)
    a.keep_logits = b.keep_logits = False
    a.tail_rows, b.tail_rows = True, False
    sa, sb = run_text(a, batch), run_text(b, batch)
    assert "xout_tail" in sa.dev and "xout_tail" not in sb.dev
    torch.testing.assert_close(sa.dev["loss_out"], sb.dev["loss_out"], rtol=1e-6, atol=0)
    assert float((a.proj.g - b.proj.g).norm() / b.proj.g.norm()) < 5e-3


def test_residual_adds_in_the_norm_kernels_change_nothing(mid):
    """TasuModel.resid_in_norm on the CPU double: o / down write bf16, the next norm adds them to the fp32 stream -- the same
    rounding points as the projections' residual mode, hence the same loss and gradients exactly."""
    geo, sd = mid
    batch = synthetic_text_batch(geo, 3, seed=12, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=12,
                                 noise=True, drop_prob=0.1, ragged=True)
    a, b = build(geo, sd), build(geo, sd)
    a.keep_logits = b.keep_logits = False
    a.resid_in_norm, b.resid_in_norm = True, False
    sa, sb = run_text(a, batch), run_text(b, batch)
    assert torch.equal(sa.dev["loss_out"], sb.dev["loss_out"])
    assert torch.equal(a.proj.g, b.proj.g)


@pytest.mark.parametrize("quantise", [False, True])
@pytest.mark.parametrize("lpw,min_len", [(1.0, 1), (2.0, 1), (0.5, 4), (1.0, 6)])
@pytest.mark.parametrize("nb", [1, 2, 3, 4])
def test_device_beam_update_restatement_matches_host_beam_state(nb, lpw, min_len, quantise):
    """The scalar restatement of the device beam-update kernel (tests/fake_ops.py::beam_update over back-pointers, the double
    the HIP kernel is compared with bit for bit) against the vectorised BeamState (pinned against the oracle's loop above):
    identical tokens on synthetic score streams with EOS events and, quantised, exact score ties."""
    from beam_stream import make_table, run_device, run_host
    B, V, T, eos = 3, 50, 12, 7
    table = make_table(0, V, eos, quantise)
    want = run_host(table, B, nb, T, eos, lpw, min_len)
    got, _ = run_device(FakeOps(), "cpu", table, B, nb, T, eos, lpw, min_len, extra_steps=2)
    assert np.array_equal(got, want), (got, want)
    assert (got == eos).any()


@pytest.mark.parametrize("k", [1, 2])
def test_linear_projector_step_matches_oracle_and_reference(k):
    """The alternate projector (``encoder_projector="linear"``: k frames concatenated -> Linear -> ReLU -> Linear, no norm; flat
    bucket [W1 | b1 | W2 | b2]) through the product's host code on the CPU double: loss / gradients against the bf16 oracle,
    loss against the REAL reference's fp32 golden, state-dict round trip under the reference's key names, and the gradient
    exchange ranges tiling the bucket."""
    from conftest import linear_projector_case
    geo, sd, batch, z = linear_projector_case(k)
    model = build(geo, sd)
    assert model.proj.names == ("linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias") and model.proj.k == k
    st = run_text(model, batch)
    out, grads = O.loss_and_projector_grads(sd, batch, dataclasses.asdict(geo), "bf16")
    assert abs(float(st.dev["loss_out"][0]) - float(out["loss"])) < 2e-3
    assert abs(float(st.dev["loss_out"][0]) - float(z["loss"])) < 2e-2
    mine = model.projector_grads()
    assert sorted(mine) == sorted(grads)
    for name, g in grads.items():
        assert mine[name].shape == g.shape, name
        assert float(torch.nn.functional.cosine_similarity(mine[name].flatten(), g.flatten(), dim=0)) > 0.999, name
    for name, v in model.projector_state_dict().items():
        assert torch.equal(v, sd[name]), name
    for chunks in (1, 4):
        ranges = sorted(model.grad_ranges(chunks))
        assert ranges[0][0] == 0 and ranges[-1][1] == model.proj.numel and len(ranges) == chunks + 1
        assert all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))


@pytest.mark.parametrize("k", [1, 2])
def test_cov1d_projector_step_matches_oracle_and_reference(k):
    """``encoder_projector="cov1d-linear"`` (Conv1d kernel = stride = k as one GEMM over k concatenated frames -> ReLU -> Linear ->
    ReLU -> Linear; flat bucket [conv W | conv b | W1 | b1 | W2 | b2]) through the product's host code on the CPU double: loss /
    gradients against the bf16 oracle, loss against the REAL reference's fp32 golden, state-dict round trip under the reference's
    key names (conv1d.weight in its [out, in, tap] layout), and the gradient exchange ranges tiling the bucket."""
    from conftest import cov1d_projector_case
    geo, sd, batch, z = cov1d_projector_case(k)
    model = build(geo, sd)
    assert model.proj.names[:2] == ("conv1d.weight", "conv1d.bias") and model.proj.k == k and model.proj.kin == 1
    st = run_text(model, batch)
    out, grads = O.loss_and_projector_grads(sd, batch, dataclasses.asdict(geo), "bf16")
    assert abs(float(st.dev["loss_out"][0]) - float(out["loss"])) < 2e-3
    assert abs(float(st.dev["loss_out"][0]) - float(z["loss"])) < 2e-2
    mine = model.projector_grads()
    assert sorted(mine) == sorted(grads)
    for name, g in grads.items():
        assert mine[name].shape == g.shape, name
        assert float(torch.nn.functional.cosine_similarity(mine[name].flatten(), g.flatten(), dim=0)) > 0.999, name
    for name, v in model.projector_state_dict().items():
        assert torch.equal(v, sd[name]), name
    for chunks in (1, 4):
        ranges = sorted(model.grad_ranges(chunks))
        assert ranges[0][0] == 0 and ranges[-1][1] == model.proj.numel and len(ranges) == chunks + 2
        assert all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))


def test_cross_attention_projector_step_matches_oracle_and_reference():
    """``encoder_projector="cross-attention"`` (W_q, then per head two GEMMs against the LLM's embedding table around the
    scale + softmax row kernel) through the product's host code on the CPU double: loss / W_q gradient against the bf16 oracle,
    loss against the REAL reference's fp32 golden, state-dict round trip, one exchange range."""
    from conftest import ca_projector_case
    geo, sd, batch, z = ca_projector_case()
    model = build(geo, sd)
    assert model.proj.names == ("W_q.weight",) and model.proj.is_ca
    st = run_text(model, batch)
    out, grads = O.loss_and_projector_grads(sd, batch, dataclasses.asdict(geo), "bf16")
    assert abs(float(st.dev["loss_out"][0]) - float(out["loss"])) < 2e-3
    assert abs(float(st.dev["loss_out"][0]) - float(z["loss"])) < 2e-2
    mine = model.projector_grads()
    assert sorted(mine) == sorted(grads) == ["encoder_projector.W_q.weight"]
    g, m = grads["encoder_projector.W_q.weight"], mine["encoder_projector.W_q.weight"]
    assert m.shape == g.shape and float(torch.nn.functional.cosine_similarity(m.flatten(), g.flatten(), dim=0)) > 0.995
    assert torch.equal(model.projector_state_dict()["encoder_projector.W_q.weight"], sd["encoder_projector.W_q.weight"])
    assert model.grad_ranges(4) == [(0, model.proj.numel)]


def test_shape_buckets_pad_without_changing_the_step(mid):
    """TasuModel.shape_buckets (graph reuse for real data): the batch is padded to the next multiple of (token columns, posterior
    rows, labelled rows) with masked columns / zero rows / ignored labels; loss, accuracy, count and gradients are those of the
    unpadded batch, and different raw shapes inside one bucket give the same step shapes."""
    geo, sd = mid
    plain, bucketed = build(geo, sd), build(geo, sd)
    bucketed.shape_buckets = (16, 8, 256)
    keys = set()
    for seed, n_audio, tl in ((3, 21, 17), (4, 19, 17), (5, 20, 18)):
        batch = synthetic_text_batch(geo, 3, seed=seed, prompt_len=9, n_audio=n_audio, target_len=tl, speech_pos=4, feat_frames=12,
                                     noise=True, drop_prob=0.0, ragged=True)
        sp, sb = run_text(plain, batch), run_text(bucketed, batch)
        assert sb.S % 16 == 0 and sb.Ra % (3 * 8) == 0 and sb.nLp % 256 == 0 and sb.S >= sp.S and sb.S - sp.S < 16
        keys.add(bucketed._shape_key(sb, "fwd"))
        torch.testing.assert_close(sp.dev["loss_out"][:3], sb.dev["loss_out"][:3], rtol=1e-6, atol=1e-7)
        assert float((plain.proj.g - bucketed.proj.g).norm() / plain.proj.g.norm()) < 5e-3
    assert len(keys) == 1, keys


def test_effective_min_length_is_hfs_under_inputs_embeds():
    """HF counts ``min_length`` including the prompt and subtracts the embedded prompt's length when the prompt arrives as
    ``inputs_embeds`` (GenerationMixin._prepare_generated_length; swept against the real reference in DESIGN.md 5): EOS is banned for
    max(min_length - S, 0) generated positions.  The product's decode entry points and the oracle apply the same rule."""
    from ps_slm_amd.decode import BeamState, effective_min_length
    assert [effective_min_length(ml, 20) for ml in (1, 6, 20, 21, 22, 24, 27)] == [0, 0, 0, 1, 2, 4, 7]
    assert effective_min_length(6, 0) == 6 and effective_min_length(0, 5) == 0
    bs = BeamState(1, 2, 8, eos=9, pad=9, min_length=effective_min_length(6, 20))
    assert not bs.ban_eos()                                                    # the reference's default (and any min_length <= S): never banned
    bs = BeamState(1, 2, 8, eos=9, pad=9, min_length=effective_min_length(23, 20))
    assert bs.ban_eos() and bs.min_length == 3
