"""TEST INFRASTRUCTURE ONLY -- tests/golden/mid512_text_ca.npz: text-only forward/backward of the REAL reference with the alternate
projector ``encoder_projector="cross-attention"`` (EncoderProjectorCTCCA, Multitask/model/projector.py:104-126: Q = W_q(posterior),
8 heads attending over every row of the LLM's embedding table; selected at Multitask/model/ps-slm.py:214-215, called at :475-480)
at a kernel-compatible geometry whose head width is a multiple of 64 (llm_dim 512 -> 8 heads of 64; the "mid" geometry's 256 / 8 =
32 is below the GEMM kernels' K granule).  Weights come from ps_slm_amd.synthetic.random_state_dict (seeded); only the seeds
and the reference's outputs are stored.  Run in the build container only:  python oracle/make_golden_ca.py"""
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.make_golden import run_fwd_bwd, save  # noqa: E402
from oracle.ref_import import build_reference_model  # noqa: E402

MID512 = dict(llm_dim=512, llm_heads=4, llm_kv_heads=2, llm_inter=1024)


def main():
    from ps_slm_amd.model import Geometry
    from ps_slm_amd.synthetic import MID_GEOMETRY, random_state_dict, synthetic_text_batch

    geo = Geometry.from_dict(dict(MID_GEOMETRY, projector="cross-attention", **MID512))
    gd = dataclasses.asdict(geo)
    seed_w, seed_b = 5151, 47
    sd = random_state_dict(geo, seed_w, with_encoder=False)
    model = build_reference_model(gd, 0, dict(gt_emb=True, gt_emb_noise=False), projector="cross-attention")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.startswith("encoder.") or m == "llm.lm_head.weight" for m in missing), (missing, unexpected)
    batch = synthetic_text_batch(geo, 3, seed=seed_b, prompt_len=9, n_audio=22, target_len=17, speech_pos=4, feat_frames=12,
                                 noise=False, ragged=True)
    GT = [" ".join(map(str, p)) for p in batch["post_ids"]]
    r = run_fwd_bwd(model, batch, GT, batch["input_features"], batch["input_feature_length"])
    g = torch.Generator().manual_seed(5)
    cols = torch.randperm(geo.llm_vocab, generator=g)[:64].sort().values
    lg = r.pop("logits")
    save("mid512_text_ca", seed_w=seed_w, seed_b=seed_b, loss=r["loss"], acc=r["acc"], cols=cols, logits_cols=lg[:, :, cols],
         lse=torch.logsumexp(lg, -1), **{"grad.W_q.weight": r["grad.W_q.weight"]})
    print(f"loss {float(r['loss']):.5f} acc {float(r['acc']):.4f} S {lg.shape[1]} |grad| {float(r['grad.W_q.weight'].norm()):.4e}")


if __name__ == "__main__":
    main()
