"""TEST INFRASTRUCTURE ONLY -- tests/golden/mid_text_cov1d_k{1,2}.npz: text-only forward/backward of the REAL reference with
the alternate projector ``encoder_projector="cov1d-linear"`` (EncoderProjectorCov1d, Multitask/model/projector.py:53-73:
Conv1d(203, 203, kernel = stride = k) -> ReLU -> Linear(203, 2048) -> ReLU -> Linear(2048, 256); the class fixes the hidden
width at 2048) at the kernel-compatible mid geometry, encoder_projector_ds_rate = 1 and 2.  Weights come from
ps_slm_amd.synthetic.random_state_dict (seeded); only the seeds and the reference's outputs are stored.
Run in the build container only:  python oracle/make_golden_cov1d.py"""
import dataclasses
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.make_golden import quiet, run_fwd_bwd, save  # noqa: E402
from oracle.ref_import import build_reference_model  # noqa: E402


def main():
    from ps_slm_amd.model import Geometry
    from ps_slm_amd.synthetic import MID_GEOMETRY, random_state_dict, synthetic_text_batch

    for k in (1, 2):
        geo = Geometry.from_dict(dict(MID_GEOMETRY, projector="cov1d-linear", projector_ds_rate=k, bottleneck=2048))
        gd = dataclasses.asdict(geo)
        seed_w, seed_b = 4040 + k, 43
        sd = random_state_dict(geo, seed_w, with_encoder=False)
        model = build_reference_model(gd, 0, dict(gt_emb=True, gt_emb_noise=False), projector="cov1d-linear", ds_rate=k)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not unexpected and all(m.startswith("encoder.") or m == "llm.lm_head.weight" for m in missing), (missing, unexpected)
        batch = synthetic_text_batch(geo, 3, seed=seed_b, prompt_len=9, n_audio=22, target_len=17, speech_pos=4, feat_frames=12,
                                     noise=False, ragged=True)
        GT = [" ".join(map(str, p)) for p in batch["post_ids"]]
        r = run_fwd_bwd(model, batch, GT, batch["input_features"], batch["input_feature_length"])
        g = torch.Generator().manual_seed(5)
        cols = torch.randperm(geo.llm_vocab, generator=g)[:64].sort().values
        lg = r.pop("logits")
        save(f"mid_text_cov1d_k{k}", seed_w=seed_w, seed_b=seed_b, k=k, loss=r["loss"], acc=r["acc"], cols=cols,
             logits_cols=lg[:, :, cols], lse=torch.logsumexp(lg, -1),
             **{"grad.conv1d.bias": r["grad.conv1d.bias"], "grad.conv1d.weight.rows8": r["grad.conv1d.weight"][::8],
                "grad.linear1.bias": r["grad.linear1.bias"], "grad.linear2.bias": r["grad.linear2.bias"],
                "grad.linear2.weight.rows16": r["grad.linear2.weight"][::16],
                "grad.linear1.weight.rows64": r["grad.linear1.weight"][::64]})
        print(f"k={k}: loss {float(r['loss']):.5f} acc {float(r['acc']):.4f} S {lg.shape[1]}")


if __name__ == "__main__":
    main()
