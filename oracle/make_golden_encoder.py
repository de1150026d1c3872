"""TEST INFRASTRUCTURE ONLY -- tests/golden/encoder_random.npz: 8 random ragged batches through the REAL reference's SANM
encoder + CTC head (Multitask/model/SenseVoice.py: SenseVoiceEncoderSmall.forward and ctc.ctc_lo) at the tiny geometry with
the weights of tests/golden/weights_tiny.npz: batch 1-4, 3-25 frames (shorter than, equal to and longer than the FSMN
kernel of 11), ragged lengths with the longest row always full (sequence_mask() uses lengths.max()).
Run in the build container only:  python oracle/make_golden_encoder.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.make_golden import GEO, SEED, quiet  # noqa: E402
from oracle.ref_import import build_reference_model  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "encoder_random.npz")


def main():
    model = quiet(build_reference_model, GEO, SEED, dict(gt_emb=True, gt_emb_noise=False))
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights_tiny.npz"))
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(w[k]) for k in w.files}, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    model.eval()
    rng = np.random.default_rng(314)
    arrs, n = {}, 0
    for case in range(8):
        B, T = int(rng.integers(1, 5)), int(rng.choice([3, 7, 11, 12, 18, 25]))
        speech = torch.from_numpy(rng.standard_normal((B, T, GEO["feat_dim"])).astype(np.float32))
        lens = rng.integers(1, T + 1, B)
        lens[int(rng.integers(0, B))] = T
        slen = torch.from_numpy(lens)
        with torch.no_grad():
            enc_out, olens = model.encoder.encoder(speech.clone(), slen)
            ctc = torch.softmax(model.encoder.ctc.ctc_lo(enc_out), dim=-1)
        arrs.update({f"c{n}_speech": speech.numpy(), f"c{n}_speech_lengths": lens, f"c{n}_enc_out": enc_out.numpy(),
                     f"c{n}_olens": olens.numpy(), f"c{n}_ctc_posterior": ctc.numpy()})
        n += 1
    arrs["n_cases"] = np.asarray(n)
    np.savez_compressed(OUT, **arrs)
    print(n, "cases,", f"{os.path.getsize(OUT) / 1024:.1f} KB")


if __name__ == "__main__":
    main()
