"""TEST INFRASTRUCTURE ONLY -- tests/golden/merge_random.npz: 48 random ragged batches pushed through the REAL reference's
``_merge_input_ids_with_audio_features`` (Multitask/model/ps-slm.py:679-873, model built by oracle/ref_import.py at the tiny
geometry).  Token embeddings and audio features are synthetic tags (token id t -> -(t + 1), audio frame (b, j) ->
1000 * (b + 1) + j), so the merged embedding tells, per output position, where the reference took it from.
Run in the build container only:  python oracle/make_golden_merge.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.make_golden import GEO, SEED, quiet  # noqa: E402
from oracle.ref_import import build_reference_model  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "merge_random.npz")


def main():
    model = quiet(build_reference_model, GEO, SEED, dict(gt_emb=True, gt_emb_noise=False))
    rng = np.random.default_rng(20260109)
    sp, eos = GEO["speech_id"], GEO["eos_id"]
    arrs = {}
    n = 0
    for case in range(48):
        B = int(rng.integers(1, 6))
        side = "left" if case % 3 == 2 else "right"
        with_labels = case % 4 != 3
        rows = []
        for _ in range(B):
            pre, post, tgt = int(rng.integers(0, 5)), int(rng.integers(0, 4)), int(rng.integers(1, 7))
            ids = rng.integers(0, 270, pre).tolist() + [sp] + rng.integers(0, 270, post).tolist()
            lab = [-100] * len(ids)
            if with_labels:
                t = rng.integers(0, 270, tgt).tolist()
                ids, lab = ids + t, lab + t
            rows.append((ids, lab))
        L = max(len(r[0]) for r in rows)
        ids_t, am_t, lab_t = [], [], []
        for ids, lab in rows:
            k = L - len(ids)
            if side == "right":
                ids_t.append(ids + [eos] * k), am_t.append([1] * len(ids) + [0] * k), lab_t.append(lab + [-100] * k)
            else:
                ids_t.append([eos] * k + ids), am_t.append([0] * k + [1] * len(ids)), lab_t.append([-100] * k + lab)
        ids_t, am_t, lab_t = torch.tensor(ids_t), torch.tensor(am_t).bool(), torch.tensor(lab_t)
        na = torch.tensor(rng.integers(1, 9, B))
        Lmax = int(na.max())
        audio = torch.zeros(B, Lmax, 4)
        for b in range(B):
            audio[b, :, :] = (1000.0 * (b + 1) + torch.arange(Lmax))[:, None]
        tok = -(ids_t.float() + 1.0)[:, :, None].expand(-1, -1, 4).contiguous()
        emb, fmask, flab, fpos, _ = model._merge_input_ids_with_audio_features(
            audio, na, tok, ids_t, am_t, lab_t if with_labels else None)
        tag = emb[:, :, 0]
        arrs.update({f"c{n}_input_ids": ids_t.numpy(), f"c{n}_attention_mask": am_t.numpy(), f"c{n}_num_audio": na.numpy(),
                     f"c{n}_source_tag": tag.numpy().astype(np.float32), f"c{n}_mask": fmask.numpy(),
                     f"c{n}_position_ids": fpos.numpy()})
        if with_labels:
            arrs[f"c{n}_labels"] = lab_t.numpy()
            arrs[f"c{n}_merged_labels"] = flab.numpy()
        n += 1
    arrs["n_cases"] = np.asarray(n)
    np.savez_compressed(OUT, **arrs)
    print(n, "cases,", f"{os.path.getsize(OUT) / 1024:.1f} KB")


if __name__ == "__main__":
    main()
