"""TEST INFRASTRUCTURE ONLY -- tests/golden/mid_audio_raw_*.npz: the REAL reference's raw-feature branch
(train_config.ctc_posterior=false, Multitask/model/ps-slm.py:515-523: PSD's decisions from the CTC posterior, the rows it keeps /
averages from the encoder's output states, those into the projector) at the kernel-compatible mid geometry, fp32, one
forward + backward, for ``encoder_projector=linear`` with k = 2 frames per row and ``linear-silu`` (k = 1).  Encoder, CTC head,
features and batch are those of mid_audio_psd (PSD lengths stable under bf16 rounding); only seeds and outputs are stored.
Run in the build container only:  python oracle/make_golden_raw.py"""
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle.make_golden import quiet, run_fwd_bwd, save  # noqa: E402
from oracle.ref_import import build_reference_model  # noqa: E402


def main():
    from conftest import mid_audio_psd_case
    from ps_slm_amd.synthetic import random_state_dict

    geo0, sd0, batch, zp = mid_audio_psd_case()
    GT = ["0"] * 3
    for kind, k, seed_p in (("linear", 2, 5151), ("linear-silu", 1, 5152)):
        geo = dataclasses.replace(geo0, projector=kind, projector_ds_rate=k, proj_in=geo0.enc_dim,
                                  bottleneck=2048 if kind == "linear" else geo0.bottleneck)
        gd = dataclasses.asdict(geo)
        sd = {n: v for n, v in sd0.items() if not n.startswith("encoder_projector.")}
        sd.update({n: v for n, v in random_state_dict(geo, seed_p, with_encoder=False).items() if n.startswith("encoder_projector.")})
        gd_ref = dict(gd, ctc_vocab=geo.ctc_vocab)
        # the reference sizes its projector from model_config.encoder_dim: the encoder's output width in this branch
        model = build_reference_model(dict(gd_ref), 0, dict(gt_emb=False, gt_emb_noise=False, ctc_posterior=False, do_psd=True),
                                      projector=kind, ds_rate=k, projector_in=geo.enc_dim)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not unexpected and set(missing) <= {"llm.lm_head.weight"}, (missing, unexpected)
        r = run_fwd_bwd(model, batch, GT, batch["input_features"], batch["input_feature_length"])
        with torch.no_grad():
            q = model.encoder.embed(torch.tensor([[0, 1, 2, 2]])).repeat(3, 1, 1)
            eo, ol = model.encoder.encoder(torch.cat([q, batch["input_features"]], 1), batch["input_feature_length"] + 4)
            cp = torch.softmax(model.encoder.ctc.ctc_lo(eo), -1)[:, 4:]
            po, pl = quiet(model.psd, eo[:, 4:], torch.clamp(ol - 4, min=0), cp, 0)
        assert torch.equal(pl, torch.from_numpy(zp["psd_lens"]))
        g = torch.Generator().manual_seed(5)
        cols = torch.randperm(geo.llm_vocab, generator=g)[:64].sort().values
        lg = r.pop("logits")
        save(f"mid_audio_raw_{kind}_k{k}", seed_p=seed_p, k=k, psd_lens=pl, psd_rows=po[:, ::3, ::8], loss=r["loss"], acc=r["acc"], cols=cols,
             logits_cols=lg[:, :, cols], lse=torch.logsumexp(lg, -1),
             **{k2: (v[::4, ::4] if v.dim() == 2 and v.numel() > 65536 else v) for k2, v in r.items() if k2.startswith("grad.")})   # large gradients: every 4th row / column
        print(kind, k, "loss", float(r["loss"]), "psd lens", pl.tolist(), "S", lg.shape[1])


if __name__ == "__main__":
    main()
