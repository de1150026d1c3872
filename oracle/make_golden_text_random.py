"""TEST INFRASTRUCTURE ONLY -- tests/golden/text_random.npz: 8 random text-only batches (1-4 utterances, ragged prompts /
targets / pseudo-posterior lengths, right and left padding) through the REAL reference's forward + backward at the tiny
geometry with the weights of tests/golden/weights_tiny.npz: loss, accuracy, the logits and the six projector gradients.
Run in the build container only:  python oracle/make_golden_text_random.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.make_golden import GEO, SEED, quiet, run_fwd_bwd, text_batch  # noqa: E402
from oracle.ref_import import build_reference_model  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "text_random.npz")


def main():
    model = quiet(build_reference_model, GEO, SEED, dict(gt_emb=True, gt_emb_noise=False))
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights_tiny.npz"))
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(w[k]) for k in w.files}, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    rng = np.random.default_rng(2718)
    sp, eos = GEO["speech_id"], GEO["eos_id"]
    arrs, n = {}, 0
    for case in range(8):
        B = int(rng.integers(1, 5))
        rows = [(rng.integers(0, 270, int(rng.integers(0, 5))).tolist() + [sp] + rng.integers(0, 270, int(rng.integers(0, 4))).tolist(),
                 rng.integers(0, 270, int(rng.integers(1, 9))).tolist() + [eos]) for _ in range(B)]
        batch = text_batch(rows, "left" if case % 3 == 2 else "right")
        post_ids = [rng.integers(1, GEO["ctc_vocab"], int(rng.integers(1, 12))).tolist() for _ in range(B)]
        GT = [" ".join(map(str, p)) for p in post_ids]
        r = run_fwd_bwd(model, batch, GT, torch.zeros(B, 8, GEO["feat_dim"]), torch.full((B,), 8))
        arrs.update({f"c{n}_{k}": (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in {**batch, **r}.items()})
        arrs[f"c{n}_post_ids_flat"] = np.concatenate([np.asarray(p) for p in post_ids])
        arrs[f"c{n}_post_lens"] = np.asarray([len(p) for p in post_ids])
        n += 1
    arrs["n_cases"] = np.asarray(n)
    np.savez_compressed(OUT, **arrs)
    print(n, "cases,", f"{os.path.getsize(OUT) / 1024:.1f} KB")


if __name__ == "__main__":
    main()
