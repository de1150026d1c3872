"""TEST INFRASTRUCTURE ONLY -- tests/golden/psd_random.npz: 32 random batches of peaky CTC-like posteriors through the REAL
reference's ``psd`` (Multitask/model/ps-slm.py:237-317; model built by oracle/ref_import.py): repeated ids, blank runs,
blank probabilities on both sides of the 0.90 threshold, ragged lengths including 0.
Run in the build container only:  python oracle/make_golden_psd.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.make_golden import GEO, SEED, quiet  # noqa: E402
from oracle.ref_import import build_reference_model  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "psd_random.npz")


def main():
    model = quiet(build_reference_model, GEO, SEED, dict(gt_emb=True, gt_emb_noise=False))
    rng = np.random.default_rng(4242)
    arrs, n = {}, 0
    for case in range(32):
        B, T, V = int(rng.integers(1, 5)), int(rng.integers(4, 41)), int(rng.integers(6, 40))
        post = np.zeros((B, T, V), dtype=np.float32)
        for b in range(B):
            t = 0
            while t < T:
                run = int(rng.integers(1, 8))
                tok = 0 if rng.random() < 0.45 else int(rng.integers(1, V))
                for _ in range(min(run, T - t)):
                    peak = float(rng.choice([0.55, 0.8, 0.89, 0.91, 0.97, 0.999]))
                    rest = rng.random(V).astype(np.float32) + 1e-3
                    rest[tok] = 0.0
                    rest *= (1.0 - peak) / rest.sum()
                    if peak < 0.6 and tok != 0:
                        rest[0] = 0.0                      # keep the argmax unambiguous
                        rest *= (1.0 - peak) / max(rest.sum(), 1e-9)
                    post[b, t] = rest
                    post[b, t, tok] = peak
                    t += 1
        lens = rng.integers(0 if case % 5 == 4 else 1, T + 1, B)
        lens[int(rng.integers(0, B))] = T
        pt, lt = torch.from_numpy(post), torch.from_numpy(lens)
        with torch.no_grad():
            out, nl = quiet(model.psd, pt, lt, pt, 0)
        arrs.update({f"c{n}_posterior": post, f"c{n}_lens": lens, f"c{n}_out": out.numpy(), f"c{n}_new_lens": nl.numpy()})
        n += 1
    arrs["n_cases"] = np.asarray(n)
    np.savez_compressed(OUT, **arrs)
    print(n, "cases,", f"{os.path.getsize(OUT) / 1024:.1f} KB")


if __name__ == "__main__":
    main()
