"""TEST INFRASTRUCTURE ONLY -- tests/golden/mid_generate_fp32.npz: 24 UNFILTERED random decode cases at the kernel-compatible "mid"
geometry with the REAL reference's ``generate`` tokens (fp32, as Multitask/inference_batch.py:113-117 runs it).

oracle/make_golden_generate_margin.py keeps only cases whose beam-search decisions survive bf16 rounding noise, because the bf16
decode path can be compared token for token on nothing else.  The fp32 decode path (ps_slm_amd/decode_fp32.py, round 6) computes
what the reference computes, so it is pinned on cases drawn WITHOUT that selection: every prompt the seeded generator draws is
kept, whether or not a bf16 evaluation of the same network decodes it the same way.  ``bf16_oracle_agrees`` records, per case,
whether the bf16-mode oracle reproduces the reference's tokens (evidence that the set contains rounding-sensitive cases; the
tests print the count).  The fp32-mode oracle must reproduce every case (checked here and in tests/test_oracle_golden.py).
Only prompts and tokens are stored; the weights are regenerated from the seed (ps_slm_amd.synthetic.decode_fixture_state_dict).
Run in the build container only:  python oracle/make_golden_generate_fp32.py"""
import dataclasses
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import tasu_oracle as O  # noqa: E402
from oracle.make_golden import quiet  # noqa: E402
from oracle.make_golden_generate_margin import PLANS, SEED_W, make_case  # noqa: E402
from oracle.ref_import import build_reference_model  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "mid_generate_fp32.npz")
N_CASES = 24
EXTRA_PLANS = [dict(num_beams=4, max_new_tokens=30, min_length=20), dict(num_beams=4, max_new_tokens=24), dict(num_beams=3, max_new_tokens=20, min_length=12)]


def main():
    from ps_slm_amd.model import Geometry
    from ps_slm_amd.synthetic import MID_GEOMETRY, decode_fixture_state_dict

    torch.set_num_threads(4)
    geo = Geometry.from_dict(MID_GEOMETRY)
    gd = dataclasses.asdict(geo)
    sd = decode_fixture_state_dict(geo, SEED_W)
    model = build_reference_model(gd, 0, dict(gt_emb=True, gt_emb_noise=False))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("encoder.") or k == "llm.lm_head.weight" for k in missing), (missing, unexpected)
    model.eval()
    plans = PLANS + EXTRA_PLANS
    arrs, agree = {}, []
    for n in range(N_CASES):
        kw = plans[n % len(plans)]
        nb, new = kw.get("num_beams", 4), kw["max_new_tokens"]
        okw = dict(num_beams=nb, max_new_tokens=new, min_length=kw.get("min_length", 1), length_penalty=kw.get("length_penalty", 1.0))
        seed = 77000 + n                                              # ONE draw per case: nothing is rejected
        rng = np.random.default_rng(seed)
        ids, am, targets = make_case(geo, rng, 3)
        post_ids = [model.encoder_tokenizer.encode(t) for t in targets]
        with torch.no_grad():
            toks = quiet(model.generate, input_ids=ids, input_features=torch.zeros(len(post_ids), 8, geo.feat_dim), attention_mask=am,
                         input_feature_length=torch.full((len(post_ids),), 8), targets=targets, **kw)
        post, plen = O.pseudo_posterior(post_ids, geo.ctc_vocab)
        outs = {}
        for mode in ("fp32", "bf16"):
            emb, mask, _, _ = O.merge(O.projector(sd, post, mode), plen, sd["llm.model.embed_tokens.weight"][ids], ids, am, None, geo.speech_id)
            outs[mode] = O.beam_search_generate(sd, emb.detach(), mask, gd, mode=mode, **okw)
        # the fp32-mode oracle restates the reference's arithmetic (same torch operators, no KV cache): it must decode every case
        assert outs["fp32"].shape == toks.shape and torch.equal(outs["fp32"], toks), (n, outs["fp32"], toks)
        agree.append(bool(outs["bf16"].shape == toks.shape and torch.equal(outs["bf16"], toks)))
        arrs.update({f"c{n}_input_ids": ids.numpy(), f"c{n}_attention_mask": am.numpy(), f"c{n}_tokens": toks.numpy(),
                     f"c{n}_post_ids_flat": np.concatenate([np.asarray(p) for p in post_ids]),
                     f"c{n}_post_lens": np.asarray([len(p) for p in post_ids]), f"c{n}_kw": np.asarray([nb, new, kw.get("min_length", 1)]),
                     f"c{n}_length_penalty": np.asarray(kw.get("length_penalty", 1.0)), f"c{n}_seed": np.asarray(seed)})
        print(f"case {n}: seed {seed} B={ids.shape[0]} nb={nb} new={new} bf16 oracle agrees: {agree[-1]} tokens {toks.tolist()}", flush=True)
    arrs["n_cases"] = np.asarray(N_CASES)
    arrs["bf16_oracle_agrees"] = np.asarray(agree)
    arrs["seed_w"] = np.asarray(SEED_W)
    np.savez_compressed(OUT, **arrs)
    print(N_CASES, "cases,", sum(agree), "of them also decoded identically by the bf16-mode oracle;", f"{os.path.getsize(OUT) / 1024:.1f} KB")


if __name__ == "__main__":
    main()
