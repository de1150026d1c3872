"""TEST INFRASTRUCTURE ONLY -- tests/golden/cps_noise_random.npz: 20 seeded calls of the REAL reference's
``ctc_pseudo_posterior_noise`` (Multitask/model/ps-slm.py:360-409; CPS = smoothing + random drops, drawn from torch's global
CPU generator; the last 8 calls with insert_prob > 0: duplicated / blank row insertions, :390-399).  Stored: the seed, the sentencepiece ids, drop_prob / smoothing range, and the reference's posterior + lengths.
What this pins: the ORDER and KIND of the random draws (one uniform alpha, then one rand(len) keep mask, per utterance) that
ps_slm_amd.ps_slm.slam_model_asr.draw_noise must reproduce from the same seed, and the posterior arithmetic.
Run in the build container only:  python oracle/make_golden_noise.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.make_golden import GEO, SEED, quiet  # noqa: E402
from oracle.ref_import import build_reference_model  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cps_noise_random.npz")
WORDS = ["a", "b", "c", "d", "e", "f", "g", "h", "k", "m", "p", "q", "s", "t", "u", "w"]


def main():
    model = quiet(build_reference_model, GEO, SEED, dict(gt_emb=True, gt_emb_noise=True))
    rng = np.random.default_rng(99)
    arrs, n = {}, 0
    for case in range(20):
        B = int(rng.integers(1, 5))
        texts = [" ".join(rng.choice(WORDS, int(rng.integers(1, 15))).tolist()) for _ in range(B)]
        ids = [model.encoder_tokenizer.encode(t) for t in texts]
        model.drop_prob = float(rng.choice([0.0, 0.05, 0.2, 0.5]))
        model.smooth_low, model.smooth_high = (0.0, 0.1) if case % 2 == 0 else (0.05, 0.3)
        model.insert_prob = 0.0 if case < 12 else float(rng.choice([0.2, 0.5, 1.0]))       # cases 12..: CPS insertions (:390-399)
        seed = 1000 + case
        torch.manual_seed(seed)
        post, lens = quiet(model.ctc_pseudo_posterior_noise, texts)
        if int(lens.max()) == 0:
            continue
        arrs.update({f"c{n}_seed": np.asarray(seed), f"c{n}_ids_flat": np.concatenate([np.asarray(i) for i in ids]),
                     f"c{n}_ids_lens": np.asarray([len(i) for i in ids]),
                     f"c{n}_params": np.asarray([model.drop_prob, model.smooth_low, model.smooth_high], dtype=np.float64),
                     f"c{n}_insert_prob": np.asarray(model.insert_prob),
                     f"c{n}_posterior": post.numpy(), f"c{n}_lens": lens.numpy()})
        n += 1
    arrs["n_cases"] = np.asarray(n)
    np.savez_compressed(OUT, **arrs)
    print(n, "cases,", f"{os.path.getsize(OUT) / 1024:.1f} KB")


if __name__ == "__main__":
    main()
