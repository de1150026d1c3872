"""The raw-feature branch (train_config.ctc_posterior=false, Multitask/model/ps-slm.py:515-523): PSD decides on the CTC posterior
but keeps / averages the encoder's output states, which feed the projector.  Goldens: the REAL reference (oracle/make_golden_raw.py)
at the mid geometry -- ``linear`` projector with 2 frames per row (the audio path with k > 1) and ``linear-silu``."""
import dataclasses

import numpy as np
import pytest
import torch

from conftest import mid_audio_raw_case
from fake_ops import FakeOps
from oracle import tasu_oracle as O
from ps_slm_amd.model import TasuModel

CASES = [("linear", 2), ("linear-silu", 1)]


def cosine(a, b):
    return float(torch.nn.functional.cosine_similarity(a.flatten().float().cpu(), b.flatten().float().cpu(), dim=0))


def sub(g, ref):
    """The golden keeps every 4th row / column of the large gradients."""
    return g[::4, ::4] if g.dim() == 2 and tuple(g[::4, ::4].shape) == tuple(ref.shape) and g.shape != ref.shape else g


def run_audio(model, batch):
    st = model.prepare_audio(batch["input_ids"], batch["attention_mask"], batch["labels"], batch["input_features"],
                             batch["input_feature_length"])
    model.forward_llm(st)
    model.backward(st)
    return st


def check(model, st, z, cos_min=0.99):
    assert np.array_equal(st.dev["psd_lens"], z["psd_lens"])                       # PSD lengths: exact
    res = st.dev["loss_out"].cpu()
    assert abs(float(res[0]) - float(z["loss"])) < 3e-2
    valid = torch.from_numpy(st.plan.key_mask[:, : st.S].astype(bool))
    cols = torch.from_numpy(z["cols"])
    lg = model.logits_view(st).float().cpu()
    ref = torch.from_numpy(z["logits_cols"])
    assert lg.shape[1] == ref.shape[1]
    assert float((lg[:, :, cols] - ref)[valid].abs().max() / ref[valid].abs().max()) < 3e-2
    n = 0
    for k, g in model.projector_grads().items():
        short = "grad." + k[len("encoder_projector."):]
        r = torch.from_numpy(z[short])
        assert cosine(sub(g.cpu(), r), r) > cos_min, k
        n += 1
    assert n == len([k for k in z if k.startswith("grad.")])


@pytest.mark.parametrize("kind,k", CASES)
def test_oracle_raw_branch_vs_reference(kind, k):
    geo, sd, batch, z = mid_audio_raw_case(kind, k)
    out, grads = O.loss_and_projector_grads(sd, batch, dataclasses.asdict(geo), "fp32", audio=True, raw=True)
    assert abs(float(out["loss"]) - float(z["loss"])) < 1e-4
    cols = torch.from_numpy(z["cols"])
    assert float((out["logits"][:, :, cols] - torch.from_numpy(z["logits_cols"])).abs().max()) < 2e-3
    for kk, g in grads.items():
        r = torch.from_numpy(z["grad." + kk[len("encoder_projector."):]])
        torch.testing.assert_close(sub(g, r), r, rtol=2e-3, atol=2e-5)


@pytest.mark.parametrize("kind,k", CASES)
def test_raw_branch_on_the_double_vs_reference(kind, k):
    """The product's host code (encoder -> PSD over encoder states -> k frames per projector row -> LLM, fwd + bwd) on the CPU
    double against the reference's fp32 outputs: PSD lengths exact, loss within 3e-2, logits within 3 % of the logit range,
    projector gradients cosine >= 0.99."""
    geo, sd, batch, z = mid_audio_raw_case(kind, k)
    m = TasuModel(geo, FakeOps(), "cpu")
    assert m.raw_features and m.proj.K == geo.enc_dim and m.proj.k == k
    m.load_reference_state_dict(sd)
    st = run_audio(m, batch)
    check(m, st, z)
    # the rows PSD produced are the encoder states' means (sampled in the fixture)
    Lmax = st.Fap and (int(z["psd_lens"].max()) // k) * k
    rows = st.dev["post"][: 3 * Lmax].view(3, Lmax, -1)[:, :, : geo.enc_dim]
    ref = torch.from_numpy(z["psd_rows"])                                      # [B, ceil(T_new / 3), E / 8], untruncated length
    got = rows[:, ::3, ::8]
    n = min(got.shape[1], ref.shape[1])
    assert float((got[:, :n] - ref[:, :n]).abs().max()) < 5e-2 * float(ref.abs().max())


def test_plugin_routes_ctc_posterior_false_to_the_raw_branch():
    from ps_slm_amd.config import ModelConfig, TrainConfig
    from ps_slm_amd.ps_slm import model_factory
    tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, ctc_posterior=False, do_psd=True)   # gt_emb is ignored here
    mc = ModelConfig(llm_path="synthetic:mid", encoder_projector="linear", encoder_projector_ds_rate=2, llm_dim=256, encoder_dim=256)
    model, _ = model_factory(tc, mc, device="cpu", ops=FakeOps(), init_seed=3)
    core = model.core
    assert core.raw_features and core.encoder is not None and model.gt_emb is False
    assert core.proj.K == core.geo.enc_dim == 256 and core.proj.k == 2
    assert model.state_dict()["encoder_projector.linear1.weight"].shape == (core.geo.bottleneck, 2 * 256)
    from ps_slm_amd.synthetic import synthetic_text_batch
    raw = synthetic_text_batch(core.geo, 2, seed=8, prompt_len=9, n_audio=5, target_len=11, speech_pos=4, feat_frames=14, noise=False)
    out, acc = model(input_ids=raw["input_ids"], attention_mask=raw["attention_mask"], labels=raw["labels"],
                     input_features=raw["input_features"], input_feature_length=raw["input_feature_length"], GT=None)
    assert torch.isfinite(out.loss) and model.last_state.path == "audio"
    with pytest.raises(NotImplementedError, match="cross-attention"):
        model_factory(tc, ModelConfig(llm_path="synthetic:mid", encoder_projector="cross-attention", llm_dim=256), device="cpu", ops=FakeOps())
