"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by running the REAL reference
(/root/reference, imported through oracle/ref_import.py) on seeded random-init weights at a tiny geometry.

Run in the build container only:  python oracle/make_golden.py
The fixtures are data (inputs + the reference's outputs); no reference source is stored.
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ref_import import build_reference_model  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

GEO = dict(enc_dim=16, enc_heads=2, enc_ffn=32, enc_blocks=3, enc_tp_blocks=2, enc_kernel=11, feat_dim=24,
           ctc_vocab=50, llm_vocab=300, llm_dim=32, llm_inter=64, llm_layers=2, llm_heads=4, llm_kv_heads=2,
           bottleneck=24, speech_id=290, eos_id=280, rope_theta=1e6)
SEED = 20260109


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def npify(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.detach().cpu().numpy()
        else:
            out[k] = np.asarray(v)
    return out


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **npify(arrs))
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KB")


def text_batch(rows, pad_side, with_labels=True):
    """rows: list of (prompt_ids incl. one speech id, target_ids).  Builds the collator's schema
    (Multitask/dataset/speech_dataset_large.py:240-305)."""
    seqs = []
    for prompt, target in rows:
        ids = list(prompt) + list(target)
        lab = [-100] * len(prompt) + list(target)
        seqs.append((ids, lab))
    L = max(len(s[0]) for s in seqs)
    ids_t, am_t, lab_t = [], [], []
    for ids, lab in seqs:
        n = L - len(ids)
        if pad_side == "right":
            ids_t.append(ids + [GEO["eos_id"]] * n)
            am_t.append([1] * len(ids) + [0] * n)
            lab_t.append(lab + [-100] * n)
        else:
            ids_t.append([GEO["eos_id"]] * n + ids)
            am_t.append([0] * n + [1] * len(ids))
            lab_t.append([-100] * n + lab)
    b = dict(input_ids=torch.tensor(ids_t), attention_mask=torch.tensor(am_t).bool())
    if with_labels:
        b["labels"] = torch.tensor(lab_t)
    return b


def run_fwd_bwd(model, batch, GT, feats, flen, autocast=False):
    for p in model.encoder_projector.parameters():
        p.grad = None
    ctx = torch.autocast("cpu", dtype=torch.bfloat16) if autocast else contextlib.nullcontext()
    with ctx:
        out, acc = quiet(model, input_ids=batch["input_ids"], input_features=feats,
                         attention_mask=batch["attention_mask"], input_feature_length=flen, GT=GT,
                         labels=batch["labels"])
    out.loss.backward()
    grads = {"grad." + n: p.grad.clone() for n, p in model.encoder_projector.named_parameters()}
    return dict(loss=out.loss.detach().float(), logits=out.logits.detach().float(), acc=torch.as_tensor(acc).float(),
                **grads)


def main():
    g = torch.Generator().manual_seed(SEED)
    model = build_reference_model(GEO, SEED, dict(gt_emb=True, gt_emb_noise=False))
    # break symmetric inits so every parameter matters (LayerNorm/RMSNorm weights are ones, biases zero)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    save("weights_tiny", **sd)
    save("geometry", **{k: np.asarray(v) for k, v in GEO.items()})

    V = GEO["llm_vocab"]
    sp = GEO["speech_id"]
    eos = GEO["eos_id"]

    def rnd_ids(n, hi=270):
        return torch.randint(0, hi, (n,), generator=g).tolist()

    # ---------------- case 1: text-only clean, right padding, B=3, ragged
    rows = [(rnd_ids(3) + [sp] + rnd_ids(2), rnd_ids(5) + [eos]),
            (rnd_ids(1) + [sp] + rnd_ids(1), rnd_ids(2) + [eos]),
            (rnd_ids(2) + [sp] + rnd_ids(3), rnd_ids(7) + [eos])]
    batch = text_batch(rows, "right")
    post_ids = [torch.randint(1, GEO["ctc_vocab"], (n,), generator=g).tolist() for n in (6, 3, 9)]
    GT = [" ".join(map(str, p)) for p in post_ids]
    feats = torch.randn(3, 12, GEO["feat_dim"], generator=g).half().float()
    flen = torch.tensor([12, 9, 5])
    r = run_fwd_bwd(model, batch, GT, feats, flen)
    # also the merge internals
    post, plen = model.ctc_pseudo_posterior(GT)
    proj = model.encoder_projector(post)
    tok = model.llm.get_input_embeddings()(batch["input_ids"])
    emb, fmask, flab, fpos, _ = model._merge_input_ids_with_audio_features(
        proj, plen, tok, batch["input_ids"], batch["attention_mask"], batch["labels"])
    save("text_clean_right", **batch, post_ids_flat=np.concatenate([np.asarray(p) for p in post_ids]),
         post_lens=np.asarray([len(p) for p in post_ids]), input_features=feats, input_feature_length=flen,
         merged_embeds=emb, merged_mask=fmask, merged_labels=flab, merged_position_ids=fpos, proj_out=proj, **r)
    # Same batch under CPU bf16 autocast (pins the oracle's bf16 emulation mode).  FINDING: the reference's
    # forward() raises under autocast ("Index put requires the source and destination dtypes match", at
    # ps-slm.py:867: fp32 final_embedding <- bf16 projector rows), so its use_fp16=true mode cannot run as
    # shipped.  The bf16 fixture therefore runs the reference's OWN pieces (projector, merge, llm, loss, acc)
    # under autocast, joined by the one .float() cast that line needs.
    for p in model.encoder_projector.parameters():
        p.grad = None
    with torch.autocast("cpu", dtype=torch.bfloat16):
        proj16 = model.encoder_projector(post)
    tok = model.llm.get_input_embeddings()(batch["input_ids"])
    emb16, m16, l16, p16, _ = model._merge_input_ids_with_audio_features(
        proj16.float(), plen, tok, batch["input_ids"], batch["attention_mask"], batch["labels"])
    with torch.autocast("cpu", dtype=torch.bfloat16):
        o16 = model.llm(inputs_embeds=emb16, attention_mask=m16, labels=l16, position_ids=p16)
    o16.loss.backward()
    from utils.metric import compute_accuracy as ref_acc
    acc16 = ref_acc(o16.logits.argmax(-1)[:, :-1], l16[:, 1:], ignore_label=-100)
    save("text_clean_right_bf16", loss=o16.loss.detach().float(), logits=o16.logits.detach().float(), acc=acc16,
         proj_out=proj16.detach().float(),
         **{"grad." + n: p.grad.clone() for n, p in model.encoder_projector.named_parameters()})

    # ---------------- case 2: text-only + CPS noise (alpha, keep captured by replaying the RNG)
    model.gt_emb_noise = True
    noise_seed = 777
    torch.manual_seed(noise_seed)
    alphas, keeps = [], []
    for p in post_ids:  # replay of the draw order at ps-slm.py:384,387
        alphas.append(torch.empty(()).uniform_(0.0, 0.1).item())
        keeps.append((torch.rand(len(p)) > 0.05).numpy())
    # make sure at least one token is dropped somewhere so that branch is exercised
    tries = 0
    while all(k.all() for k in keeps) and tries < 100:
        noise_seed += 1
        tries += 1
        torch.manual_seed(noise_seed)
        alphas, keeps = [], []
        for p in post_ids:
            alphas.append(torch.empty(()).uniform_(0.0, 0.1).item())
            keeps.append((torch.rand(len(p)) > 0.05).numpy())
    torch.manual_seed(noise_seed)
    r = run_fwd_bwd(model, batch, GT, feats, flen)
    save("text_noise_right", **batch, post_ids_flat=np.concatenate([np.asarray(p) for p in post_ids]),
         post_lens=np.asarray([len(p) for p in post_ids]), alphas=np.asarray(alphas, dtype=np.float64),
         keeps_flat=np.concatenate(keeps), input_features=feats, input_feature_length=flen, **r)
    model.gt_emb_noise = False

    # ---------------- case 3: left padding B=2, and B=1
    batch_l = text_batch(rows[:2], "left")
    r = run_fwd_bwd(model, batch_l, GT[:2], feats[:2], flen[:2])
    post, plen = model.ctc_pseudo_posterior(GT[:2])
    proj = model.encoder_projector(post)
    tok = model.llm.get_input_embeddings()(batch_l["input_ids"])
    emb, fmask, flab, fpos, _ = model._merge_input_ids_with_audio_features(
        proj, plen, tok, batch_l["input_ids"], batch_l["attention_mask"], batch_l["labels"])
    save("text_clean_left", **batch_l, post_ids_flat=np.concatenate([np.asarray(p) for p in post_ids[:2]]),
         post_lens=np.asarray([len(p) for p in post_ids[:2]]), input_features=feats[:2],
         input_feature_length=flen[:2], merged_mask=fmask, merged_labels=flab, merged_position_ids=fpos, **r)
    batch_1 = text_batch(rows[2:], "right")
    # (features must be padded to exactly the batch max length: sequence_mask() uses lengths.max(), SenseVoice.py:285-293)
    feats1 = feats[2:, : int(flen[2])]
    r = run_fwd_bwd(model, batch_1, GT[2:], feats1, flen[2:])
    save("text_clean_b1", **batch_1, post_ids_flat=np.asarray(post_ids[2]), post_lens=np.asarray([len(post_ids[2])]),
         input_features=feats1, input_feature_length=flen[2:], **r)

    # ---------------- case 4: encoder + CTC posterior alone (ragged lengths)
    speech = torch.randn(3, 14, GEO["feat_dim"], generator=g)
    slen = torch.tensor([14, 10, 6])
    with torch.no_grad():
        enc_out, olens = model.encoder.encoder(speech.clone(), slen)
        ctc = torch.softmax(model.encoder.ctc.ctc_lo(enc_out), dim=-1)
    save("encoder_ragged", speech=speech, speech_lengths=slen, enc_out=enc_out, olens=olens, ctc_posterior=ctc)

    # ---------------- case 5: PSD on crafted posteriors (runs > 5, all blank, L = 0, blank filter edge)
    Vc = GEO["ctc_vocab"]
    T = 16

    def craft(seq, conf=0.8, blank_leak=0.05):
        P = torch.full((T, Vc), 0.0)
        for t, c in enumerate(seq):
            noise = torch.rand(Vc, generator=g) * 0.01
            P[t] = noise
            P[t, c] += conf
            if c != 0:
                P[t, 0] += blank_leak
            P[t] /= P[t].sum()
        return P

    seqs = [[3, 3, 3, 0, 0, 7, 7, 7, 7, 7, 7, 7, 9, 0, 9, 9],      # runs, run > 5, blanks kept singly
            [0] * T,                                             # all blank -> everything filtered
            [5, 6, 6, 0, 8, 8, 2, 2, 2, 1, 1, 0, 0, 4, 4, 4],
            [1] * T]                                             # length 0 utterance (lens = 0)
    post = torch.stack([craft(s) for s in seqs])
    # one blank frame with prob below threshold (kept) and a non-blank run whose mean blank prob >= 0.9 is impossible
    post[2, 3] = 0.0
    post[2, 3, 0] = 0.6
    post[2, 3, 11] = 0.4
    lens = torch.tensor([16, 16, 13, 0])
    with torch.no_grad():
        out, nl = quiet(model.psd, post, lens, post, 0)
    save("psd_crafted", posterior=post, lens=lens, out=out, new_lens=nl)
    # log-prob input branch (ps-slm.py:256-257)
    with torch.no_grad():
        out2, nl2 = quiet(model.psd, post.clamp_min(1e-30).log(), lens, post.clamp_min(1e-30).log(), 0)
    save("psd_crafted_logprob", out=out2, new_lens=nl2)

    # ---------------- case 6: audio path fwd/bwd (encoder -> CTC -> PSD -> projector -> LLM)
    model.gt_emb = False
    with torch.no_grad():  # bias the CTC head so that runs and blanks actually occur
        model.encoder.ctc.ctc_lo.bias.zero_()
        model.encoder.ctc.ctc_lo.bias[0] = 1.5
        model.encoder.ctc.ctc_lo.bias[7] = 1.2
        model.encoder.ctc.ctc_lo.bias[9] = 1.0
        model.encoder.ctc.ctc_lo.weight.mul_(3.0)
    sd_audio = {"encoder.ctc.ctc_lo.bias": model.encoder.ctc.ctc_lo.bias.clone(),
                "encoder.ctc.ctc_lo.weight": model.encoder.ctc.ctc_lo.weight.clone()}
    feats_a = torch.randn(3, 30, GEO["feat_dim"], generator=g).half().float()
    flen_a = torch.tensor([30, 22, 17])
    r = run_fwd_bwd(model, batch, GT, feats_a, flen_a)
    with torch.no_grad():
        q = model.encoder.embed(torch.tensor([[0, 1, 2, 2]])).repeat(3, 1, 1)
        eo, ol = model.encoder.encoder(torch.cat([q, feats_a], 1), flen_a + 4)
        cp = torch.softmax(model.encoder.ctc.ctc_lo(eo), -1)[:, 4:]
        po, pl = quiet(model.psd, cp, torch.clamp(ol - 4, min=0), cp, 0)
    save("audio_psd_right", **batch, input_features=feats_a, input_feature_length=flen_a, psd_out=po, psd_lens=pl,
         **sd_audio, **r)

    # ---------------- case 7: beam-4 generate (audio path, left padding as in inference mode)
    gen_rows = [(rnd_ids(3) + [sp] + rnd_ids(2), []), (rnd_ids(1) + [sp] + rnd_ids(1), [])]
    gb = text_batch(gen_rows, "left", with_labels=False)
    with torch.no_grad():
        toks = quiet(model.generate, input_ids=gb["input_ids"], input_features=feats_a[:2],
                     attention_mask=gb["attention_mask"], input_feature_length=flen_a[:2], max_new_tokens=12)
    save("generate_audio_beam4", **gb, input_features=feats_a[:2], input_feature_length=flen_a[:2], tokens=toks,
         **sd_audio)
    model.gt_emb = True
    with torch.no_grad():
        toks = quiet(model.generate, input_ids=gb["input_ids"], input_features=feats_a[:2],
                     attention_mask=gb["attention_mask"], input_feature_length=flen_a[:2], max_new_tokens=12,
                     targets=["a b", "c d e"])
    # generate() regex-cleans `targets` to lower-case letters (ps-slm.py:592-596); the fake sentencepiece maps
    # each word to a deterministic id, recorded here so the test does not need the fake tokenizer.
    gen_ids = [model.encoder_tokenizer.encode(t) for t in ["a b", "c d e"]]
    save("generate_text_beam4", **gb, tokens=toks, post_ids_flat=np.concatenate([np.asarray(p) for p in gen_ids]),
         post_lens=np.asarray([len(p) for p in gen_ids]))


def main_mid():
    """Kernel-compatible geometry (head_dim 128, odd CTC vocabulary): weights come from
    ps_slm_amd.synthetic.random_state_dict (seeded) and are loaded into the REAL reference; only inputs'
    seeds and the reference's outputs are stored."""
    import dataclasses

    from ps_slm_amd.model import Geometry
    from ps_slm_amd.synthetic import MID_GEOMETRY, random_state_dict, synthetic_text_batch

    geo = Geometry.from_dict(MID_GEOMETRY)
    gd = dataclasses.asdict(geo)
    seed_w, seed_b = 2026, 31
    sd = random_state_dict(geo, seed_w, with_encoder=True)
    model = build_reference_model(gd, 0, dict(gt_emb=True, gt_emb_noise=False))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert set(missing) <= {"llm.lm_head.weight"}, missing
    batch = synthetic_text_batch(geo, 3, seed=seed_b, prompt_len=9, n_audio=21, target_len=17, speech_pos=4,
                                 feat_frames=12, noise=True, drop_prob=0.15, ragged=True)
    # the reference draws alpha/keep itself; feed the clean path with the already-dropped ids and emulate the
    # smoothing through the reference's own noise function by replaying its RNG order is not possible with
    # external draws, so this fixture uses the CLEAN posterior of the kept ids (alpha = 0).
    kept = [list(np.asarray(p)[np.asarray(k, dtype=bool)]) for p, k in zip(batch["post_ids"], batch["keeps"])]
    GT = [" ".join(map(str, k)) for k in kept]
    r = run_fwd_bwd(model, batch, GT, batch["input_features"], batch["input_feature_length"])
    g = torch.Generator().manual_seed(5)
    cols = torch.randperm(geo.llm_vocab, generator=g)[:64].sort().values
    lg = r.pop("logits")
    save("mid_text_clean", seed_w=seed_w, seed_b=seed_b, loss=r["loss"], acc=r["acc"], cols=cols,
         logits_cols=lg[:, :, cols], lse=torch.logsumexp(lg, -1), argmax=lg.argmax(-1),
         **{k: v for k, v in r.items() if k.startswith("grad.") and k != "grad.ffn.2.weight"},
         **{"grad.ffn.2.weight.even_rows": r["grad.ffn.2.weight"][::2]})
    # audio path at mid geometry
    model.gt_emb = False
    ra = run_fwd_bwd(model, batch, GT, batch["input_features"], batch["input_feature_length"])
    lga = ra.pop("logits")
    save("mid_audio", seed_w=seed_w, seed_b=seed_b, loss=ra["loss"], acc=ra["acc"], cols=cols,
         logits_cols=lga[:, :, cols], lse=torch.logsumexp(lga, -1), argmax=lga.argmax(-1),
         **{k: v for k, v in ra.items() if k.startswith("grad.") and k not in ("grad.ffn.2.weight", "grad.ffn.0.weight")})

    # audio path with a CTC head biased towards blank / two symbols so that PSD really merges and filters
    with torch.no_grad():
        model.encoder.ctc.ctc_lo.weight.mul_(3.0)
        model.encoder.ctc.ctc_lo.bias.zero_()
        model.encoder.ctc.ctc_lo.bias[0] = 2.5
        model.encoder.ctc.ctc_lo.bias[7] = 2.2
        model.encoder.ctc.ctc_lo.bias[9] = 2.0
    # piecewise-constant features (+ small noise) so that consecutive frames share their argmax symbol
    # PSD is discontinuous (argmax ties, the 0.90 blank threshold), so pick an input whose decisions have enough
    # margin that fp32 and bf16 arithmetic agree on the kept-frame counts (checked with the oracle's bf16 mode).
    from oracle import tasu_oracle as O
    sd_b = dict(sd)
    sd_b["encoder.ctc.ctc_lo.weight"] = model.encoder.ctc.ctc_lo.weight.detach().clone()
    sd_b["encoder.ctc.ctc_lo.bias"] = model.encoder.ctc.ctc_lo.bias.detach().clone()
    flen = torch.tensor([40, 31, 23])
    for feat_seed in range(100, 200):
        gf = torch.Generator().manual_seed(feat_seed)
        segs = torch.randn(3, 10, geo.feat_dim, generator=gf).repeat_interleave(4, dim=1)
        feats = (segs * 3.0 + 0.05 * torch.randn(3, 40, geo.feat_dim, generator=gf)).half().float()
        with torch.no_grad():
            q = model.encoder.embed(torch.tensor([[0, 1, 2, 2]])).repeat(3, 1, 1)
            eo, ol = model.encoder.encoder(torch.cat([q, feats], 1), flen + 4)
            cp = torch.softmax(model.encoder.ctc.ctc_lo(eo), -1)[:, 4:]
            po, pl = quiet(model.psd, cp, torch.clamp(ol - 4, min=0), cp, 0)
        pb, _, lb = O.audio_front(sd_b, feats, flen, geo.enc_heads, geo.enc_kernel, "bf16")
        _, plb = O.psd(pb, lb, pb, 0)
        if torch.equal(plb, pl) and int(pl.sum()) < int(flen.sum()) - 10:
            break
    else:
        raise RuntimeError("no robust PSD input found")
    print("mid_audio_psd: feature seed", feat_seed, "psd lens", pl.tolist())
    rp = run_fwd_bwd(model, batch, GT, feats, flen)
    lgp = rp.pop("logits")
    save("mid_audio_psd", seed_w=seed_w, seed_b=seed_b, input_features=feats.half(), input_feature_length=flen,
         ctc_bias=model.encoder.ctc.ctc_lo.bias.clone(), ctc_weight_scale=3.0, psd_lens=pl, loss=rp["loss"], acc=rp["acc"],
         cols=cols, logits_cols=lgp[:, :, cols], lse=torch.logsumexp(lgp, -1),
         **{k: v for k, v in rp.items() if k.startswith("grad.") and k not in ("grad.ffn.2.weight", "grad.ffn.0.weight")})

    # beam-4 generate at the mid geometry: text path (clean posterior of regex-cleaned targets) and audio path,
    # left-padded prompts as in inference mode (speech_dataset_large.py:242-245)
    gen_raw = synthetic_text_batch(geo, 2, seed=77, prompt_len=9, n_audio=21, target_len=3, speech_pos=4, feat_frames=12,
                                   noise=False, ragged=True)
    P = 9
    ids_l, am_l = [], []
    for b in range(2):
        row = gen_raw["input_ids"][b][gen_raw["attention_mask"][b]][: P - b]      # prompt only, ragged
        n = P - len(row)
        ids_l.append(torch.cat([torch.full((n,), geo.eos_id, dtype=torch.long), row]))
        am_l.append(torch.cat([torch.zeros(n, dtype=torch.bool), torch.ones(len(row), dtype=torch.bool)]))
    gids, gam = torch.stack(ids_l), torch.stack(am_l)
    model.gt_emb = True
    words = ["ab cd ef gh", "x yz"]
    with torch.no_grad():
        toks_t = quiet(model.generate, input_ids=gids, input_features=feats[:2], attention_mask=gam,
                       input_feature_length=flen[:2], max_new_tokens=16, targets=words)
    word_ids = [model.encoder_tokenizer.encode(t) for t in words]
    model.gt_emb = False
    with torch.no_grad():
        toks_a = quiet(model.generate, input_ids=gids, input_features=feats[:2], attention_mask=gam,
                       input_feature_length=flen[:2], max_new_tokens=16)
    save("mid_generate_beam4", seed_w=seed_w, input_ids=gids, attention_mask=gam, tokens_text=toks_t, tokens_audio=toks_a,
         post_ids_flat=np.concatenate([np.asarray(w) for w in word_ids]), post_lens=np.asarray([len(w) for w in word_ids]))
    print("mid_generate_beam4 text", toks_t.tolist(), "audio", toks_a.tolist())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "mid":
        main_mid()
    else:
        main()
        main_mid()
