"""TEST INFRASTRUCTURE ONLY -- tests/golden/mid_generate_lora_margin.npz: decode cases of the LoRA-ADAPTED model whose beam-search
decisions survive bf16 rounding, so that the tokens of the product's decode on the merged weights W' = bf16(W + s B A)
(ps_slm_amd/lora.py:merged_llm) are compared EXACTLY with the reference's (the un-adapted model has
tests/golden/mid_generate_margin.npz from oracle/make_golden_generate_margin.py, whose recipe this follows).

A prompt is kept when all of these produce the same tokens:
  1. the REAL reference model (oracle/ref_import.py; fp32) with the LoRA formula applied by hand to its HF decoder
     (oracle/lora_oracle.py:apply_hand_lora -- peft is not in this image; Multitask/ps-slm.py:199-216 is what it stands for),
     through its own generate() (Multitask/ps-slm.py:640-673);
  2. the bf16 oracle (oracle/tasu_oracle.py:beam_search_generate) on the state dict with W + s B A merged in fp32 from the bf16
     rounded base weights -- the arithmetic of the product's merge;
  3. N_JITTER more runs of 2 with +-1 bf16 ulp flips on the logits;
  4. the product's host code on the CPU double (tests/fake_ops.py) with the adapters enabled and loaded -- a disagreement here is
     either explained as a near-tie from the recorded logits (and the prompt rejected) or counted; the count must be 0.

Run in the build container only:  python oracle/make_golden_generate_lora_margin.py
The fixture is data: seeds, prompts and the reference's tokens; weights and adapters come from ps_slm_amd.synthetic (seeded).
"""
import dataclasses
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import tasu_oracle as O  # noqa: E402
from oracle.lora_oracle import apply_hand_lora  # noqa: E402
from oracle.make_golden import quiet  # noqa: E402
from oracle.make_golden_generate_margin import JITTER_PROB, N_JITTER, explain_disagreement, make_case  # noqa: E402
from oracle.ref_import import build_reference_model  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "mid_generate_lora_margin.npz")
SEED_W, SEED_L = 4242, 515
R, ALPHA = 16, 32
TARGETS = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")
B_SCALE = 0.05
PLANS = [dict(num_beams=4, max_new_tokens=12), dict(num_beams=4, max_new_tokens=16, length_penalty=0.7),
         dict(num_beams=2, max_new_tokens=9), dict(num_beams=4, max_new_tokens=10, min_length=6),
         dict(num_beams=1, max_new_tokens=14), dict(num_beams=3, max_new_tokens=8, length_penalty=2.0),
         dict(num_beams=4, max_new_tokens=40, min_length=34)]
PARENT = {"q_proj": "self_attn", "k_proj": "self_attn", "v_proj": "self_attn", "o_proj": "self_attn",
          "gate_proj": "mlp", "up_proj": "mlp", "down_proj": "mlp"}


def merged_state_dict(sd, lsd, geo, scaling):
    """The product's decode weights in the reference's key space, kept in fp32: bf16(W).float() + s B A (the oracle's bf16 mode
    rounds them once more, as ps_slm_amd/lora.py:merged_llm does)."""
    from ps_slm_amd.lora import key_of
    out = dict(sd)
    for l in range(geo.llm_layers):
        for t in TARGETS:
            k = f"llm.model.layers.{l}.{PARENT[t]}.{t}.weight"
            A, B = lsd[key_of(l, t, "A")].double(), lsd[key_of(l, t, "B")].double()
            out[k] = (sd[k].bfloat16().double() + scaling * (B @ A)).float()
    return out


def main():
    from fake_ops import FakeOps
    from ps_slm_amd.decode import beam_search_generate
    from ps_slm_amd.lora import LoraConfig
    from ps_slm_amd.model import Geometry, TasuModel
    from ps_slm_amd.synthetic import MID_GEOMETRY, decode_fixture_state_dict, random_lora_state_dict

    torch.set_num_threads(4)
    geo = Geometry.from_dict(MID_GEOMETRY)
    gd = dataclasses.asdict(geo)
    cfg = LoraConfig(r=R, lora_alpha=ALPHA, lora_dropout=0.0, target_modules=TARGETS)
    sd = decode_fixture_state_dict(geo, SEED_W)
    lsd = random_lora_state_dict(geo, cfg, SEED_L, b_scale=B_SCALE)
    sdm = merged_state_dict(sd, lsd, geo, cfg.scaling)
    model = build_reference_model(gd, 0, dict(gt_emb=True, gt_emb_noise=False))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("encoder.") or k == "llm.lm_head.weight" for k in missing), (missing, unexpected)
    apply_hand_lora(model.llm, lsd, TARGETS, cfg.scaling)
    model.eval()
    double = TasuModel(geo, FakeOps(), "cpu")
    double.load_reference_state_dict(sd)
    double.enable_lora(cfg)
    double.lora.load_state_dict(lsd)
    double.sync_projector_copies()
    base = TasuModel(geo, FakeOps(), "cpu")
    base.load_reference_state_dict(sd)
    arrs, n, tried, disagreements, near_ties, same_as_base = {}, 0, 0, [], [], 0
    for case, kw in enumerate(PLANS):
        nb, new = kw.get("num_beams", 4), kw["max_new_tokens"]
        okw = dict(num_beams=nb, max_new_tokens=new, min_length=kw.get("min_length", 1), length_penalty=kw.get("length_penalty", 1.0))
        seed = 5000 + 1000 * case
        while True:
            seed += 1
            tried += 1
            rng = np.random.default_rng(seed)
            ids, am, targets = make_case(geo, rng, 3 if new < 30 else 2)
            post_ids = [model.encoder_tokenizer.encode(t) for t in targets]
            post, plen = O.pseudo_posterior(post_ids, geo.ctc_vocab)
            emb, mask, _, _ = O.merge(O.projector(sd, post, "bf16"), plen, sd["llm.model.embed_tokens.weight"][ids], ids, am,
                                      None, geo.speech_id)
            emb = emb.detach()
            trace = []
            t16 = O.beam_search_generate(sdm, emb, mask, gd, mode="bf16", logits_trace=trace, **okw)
            if case % 2 == 1 and not (t16 == geo.eos_id).any():
                continue                                                     # every other case must see a beam finish early
            with torch.no_grad():
                toks = quiet(model.generate, input_ids=ids, input_features=torch.zeros(len(post_ids), 8, geo.feat_dim),
                             attention_mask=am, input_feature_length=torch.full((len(post_ids),), 8), targets=targets, **kw)
            if toks.shape != t16.shape or not torch.equal(toks, t16):
                continue
            stable = True
            for j in range(N_JITTER):
                jit = O.bf16_ulp_jitter(100 * seed + j, JITTER_PROB)
                tj = O.beam_search_generate(sdm, emb, mask, gd, mode="bf16", logits_replay=trace, logit_jitter=jit, **okw)
                if tj is None:
                    tj = O.beam_search_generate(sdm, emb, mask, gd, mode="bf16", logit_jitter=jit, **okw)
                if tj.shape != t16.shape or not torch.equal(tj, t16):
                    stable = False
                    break
            if not stable:
                continue

            def run(m):
                st = m.prepare_text(ids, am, None, post_ids, None, None)
                m.forward_projector_text(st)
                return beam_search_generate(m, st, eos_token_id=geo.eos_id, pad_token_id=geo.eos_id, **okw)

            tb = run(base)
            if tb.shape == t16.shape and torch.equal(tb, t16):
                same_as_base += 1
                continue                                                     # the adapters must change what is decoded
            tc = run(double)
            if tc.shape != t16.shape or not torch.equal(tc, t16):
                explained = explain_disagreement(double, st_factory=lambda: double.prepare_text(ids, am, None, post_ids, None, None),
                                                 trace=trace, tc=tc, t_ref=t16, okw=okw, geo=geo)
                if explained is None:
                    disagreements.append((case, seed))
                    print(f"DOUBLE DISAGREES (UNEXPLAINED) on case {case} seed {seed}: double {tc.tolist()} reference {toks.tolist()}", flush=True)
                    break
                near_ties.append((case, seed, explained))
                print(f"near-tie rejected: case {case} seed {seed}: {explained}", flush=True)
                continue
            break
        arrs.update({f"c{n}_input_ids": ids.numpy(), f"c{n}_attention_mask": am.numpy(), f"c{n}_tokens": toks.numpy(),
                     f"c{n}_tokens_base": tb.numpy(),
                     f"c{n}_post_ids_flat": np.concatenate([np.asarray(p) for p in post_ids]),
                     f"c{n}_post_lens": np.asarray([len(p) for p in post_ids]),
                     f"c{n}_kw": np.asarray([nb, new, kw.get("min_length", 1)]),
                     f"c{n}_length_penalty": np.asarray(kw.get("length_penalty", 1.0)), f"c{n}_seed": np.asarray(seed)})
        print(f"case {n}: seed {seed} B={ids.shape[0]} nb={nb} new={new} tokens {toks.tolist()}", flush=True)
        n += 1
    arrs.update(n_cases=np.asarray(n), double_disagreements=np.asarray(len(disagreements)), near_ties_rejected=np.asarray(len(near_ties)),
                prompts_tried=np.asarray(tried), same_as_base_rejected=np.asarray(same_as_base), seed_w=np.asarray(SEED_W),
                seed_l=np.asarray(SEED_L), r=np.asarray(R), alpha=np.asarray(ALPHA), b_scale=np.asarray(B_SCALE),
                targets=np.asarray(",".join(TARGETS)))
    np.savez_compressed(OUT, **arrs)
    print(n, "cases,", tried, "prompts tried,", same_as_base, "dropped because the base model decodes the same,", len(near_ties),
          "near-ties rejected,", len(disagreements), "UNEXPLAINED double disagreements;", f"{os.path.getsize(OUT) / 1024:.1f} KB")
    if disagreements:
        raise SystemExit(f"the CPU double disagrees on stable cases {disagreements}: fix the product's host code, do not drop the case")


if __name__ == "__main__":
    main()
