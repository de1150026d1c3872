"""TEST INFRASTRUCTURE ONLY -- tests/golden/generate_random.npz: 14 decode cases through the REAL reference's
``generate`` (text path: pseudo-posterior of the cleaned targets -> projector -> merge -> HF beam search) at the tiny
geometry with the weights of tests/golden/weights_tiny.npz: 1-3 utterances, left padding, 1-4 beams, different
max_new_tokens / min_length / length_penalty.  Run in the build container only:  python oracle/make_golden_generate.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.make_golden import GEO, SEED, quiet  # noqa: E402
from oracle.ref_import import build_reference_model  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "generate_random.npz")
WORDS = ["a", "b", "c", "d", "e", "f", "g", "h", "k", "m", "p", "q", "s", "t"]


def main():
    model = quiet(build_reference_model, GEO, SEED, dict(gt_emb=True, gt_emb_noise=False))
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights_tiny.npz"))
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(w[k]) for k in w.files}, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    model.eval()
    rng = np.random.default_rng(77)
    sp, eos = GEO["speech_id"], GEO["eos_id"]
    arrs, n = {}, 0
    plans = [dict(num_beams=4, max_new_tokens=12), dict(num_beams=4, max_new_tokens=5), dict(num_beams=2, max_new_tokens=9),
             dict(num_beams=3, max_new_tokens=7, length_penalty=2.0), dict(num_beams=1, max_new_tokens=8),
             dict(num_beams=4, max_new_tokens=10, min_length=6), dict(num_beams=4, max_new_tokens=6, length_penalty=0.5)]
    for case in range(14):
        kw = plans[case % len(plans)]
        B = int(rng.integers(1, 4))
        rows = [rng.integers(0, 270, int(rng.integers(1, 6))).tolist() + [sp] + rng.integers(0, 270, int(rng.integers(0, 4))).tolist()
                for _ in range(B)]
        L = max(len(r) for r in rows)
        ids = torch.tensor([[eos] * (L - len(r)) + r for r in rows])
        am = torch.tensor([[0] * (L - len(r)) + [1] * len(r) for r in rows]).bool()
        targets = [" ".join(rng.choice(WORDS, int(rng.integers(1, 6))).tolist()) for _ in range(B)]
        gen_ids = [model.encoder_tokenizer.encode(t) for t in targets]
        with torch.no_grad():
            toks = quiet(model.generate, input_ids=ids, input_features=torch.zeros(B, 8, GEO["feat_dim"]), attention_mask=am,
                         input_feature_length=torch.full((B,), 8), targets=targets, **kw)
        arrs.update({f"c{n}_input_ids": ids.numpy(), f"c{n}_attention_mask": am.numpy(), f"c{n}_tokens": toks.numpy(),
                     f"c{n}_post_ids_flat": np.concatenate([np.asarray(p) for p in gen_ids]),
                     f"c{n}_post_lens": np.asarray([len(p) for p in gen_ids]),
                     f"c{n}_kw": np.asarray([kw.get("num_beams", 4), kw.get("max_new_tokens", 200), kw.get("min_length", 1)]),
                     f"c{n}_length_penalty": np.asarray(kw.get("length_penalty", 1.0))})
        n += 1
    arrs["n_cases"] = np.asarray(n)
    np.savez_compressed(OUT, **arrs)
    print(n, "cases;", f"{os.path.getsize(OUT) / 1024:.1f} KB;", "token shapes", [arrs[f"c{i}_tokens"].shape for i in range(n)])


if __name__ == "__main__":
    main()
