"""LoRA recipe (use_peft=true), host logic on the CPU test double: the REAL host code of ps_slm_amd/lora.py + model.py driven
through tests/fake_ops.py, against goldens produced by the real reference model with the LoRA formula applied by hand
(oracle/make_golden_lora.py; peft is not available, see oracle/lora_oracle.py)."""
import numpy as np
import pytest
import torch

from conftest import load_npz
from fake_ops import FakeOps
from ps_slm_amd.lora import LoraConfig, key_of
from ps_slm_amd.model import Geometry, TasuModel
from ps_slm_amd.synthetic import MID_GEOMETRY, random_lora_state_dict, random_state_dict, synthetic_text_batch


def cosine(a, b):
    return float(torch.nn.functional.cosine_similarity(a.flatten().float().cpu(), b.flatten().float().cpu(), dim=0))


def golden_case(name):
    z = load_npz(name)
    geo = Geometry.from_dict(MID_GEOMETRY)
    cfg = LoraConfig(r=int(z["r"]), lora_alpha=float(z["alpha"]), lora_dropout=float(z["p"]),
                     target_modules=tuple(str(z["targets"]).split(",")))
    sd = random_state_dict(geo, int(z["seed_w"]), with_encoder=False)
    lsd = random_lora_state_dict(geo, cfg, int(z["seed_l"]))
    batch = synthetic_text_batch(geo, 3, seed=int(z["seed_b"]), prompt_len=9, n_audio=21, target_len=17, speech_pos=4,
                                 feat_frames=12, noise=True, drop_prob=0.15, ragged=True)
    batch["post_ids"] = [list(np.asarray(p)[np.asarray(k, dtype=bool)]) for p, k in zip(batch["post_ids"], batch["keeps"])]
    del batch["alphas"], batch["keeps"]
    return z, geo, cfg, sd, lsd, batch


def build(geo, cfg, sd, lsd, ops, device, rng=None):
    m = TasuModel(geo, ops, device)
    m.load_reference_state_dict(sd)
    m.enable_lora(cfg)
    m.lora.load_state_dict(lsd)
    m.sync_projector_copies()
    if rng is not None:
        m.lora.seed_dropout(int(rng[0]), int(rng[1]) - 1)      # the forward advances the step before it draws
    return m


def run_text(model, batch):
    st = model.prepare_text(batch["input_ids"], batch["attention_mask"], batch["labels"], batch["post_ids"])
    model.forward_projector_text(st)
    model.forward_llm(st)
    model.backward(st)
    return st


def check_against_golden(model, st, z, cos_min=0.995):
    res = st.dev["loss_out"].cpu()
    assert abs(float(res[0]) - float(z["loss"])) < 2e-2
    valid = torch.from_numpy(st.plan.key_mask[:, : st.S].astype(bool))
    cols = torch.from_numpy(z["cols"])
    lg = model.logits_view(st).float().cpu()
    ref = torch.from_numpy(z["logits_cols"])
    assert float((lg[:, :, cols] - ref)[valid].abs().max() / ref[valid].abs().max()) < 3e-2
    for k, g in model.projector_grads().items():
        short = "grad." + k[len("encoder_projector."):]
        if short in z:
            assert cosine(g, torch.from_numpy(z[short])) > cos_min, k
    lg_ = model.lora_grads()
    n = 0
    for k, g in lg_.items():
        ref = torch.from_numpy(z["lgrad." + k].astype(np.float32))
        assert g.shape == ref.shape
        assert cosine(g, ref) > cos_min, k
        assert abs(float(g.norm() / ref.norm()) - 1.0) < 5e-2, k
        n += 1
    assert n == 2 * len(model.lora.cfg.target_modules) * model.geo.llm_layers


@pytest.mark.parametrize("name", ["mid_text_lora", "mid_text_lora_qv", "mid_text_lora_drop"])
def test_lora_step_vs_reference_golden(name):
    """r = 16 (padded to the 64-wide K of the GEMMs) on all seven Linears; r = 64 on q / v only (un-adapted members of the fused
    groups); r = 64 with dropout 0.25 under a GIVEN mask.  Stated bf16 tolerances: |loss - ref| <= 2e-2, logits within 3 % of the
    logit range, every gradient cosine >= 0.995 and norm within 5 %."""
    z, geo, cfg, sd, lsd, batch = golden_case(name)
    m = build(geo, cfg, sd, lsd, FakeOps(), "cpu", rng=z["rng"] if cfg.lora_dropout > 0 else None)
    st = run_text(m, batch)
    check_against_golden(m, st, z)


def test_lora_bucket_layout_and_keys():
    geo = Geometry.from_dict(MID_GEOMETRY)
    cfg = LoraConfig(r=16, lora_alpha=32, lora_dropout=0.0)
    m = TasuModel(geo, FakeOps(), "cpu")
    m.load_reference_state_dict(random_state_dict(geo, 1, with_encoder=False))
    n_proj = m.proj.numel
    m.enable_lora(cfg)
    lp = m.lora
    assert lp.base == n_proj and m.proj.numel == n_proj + lp.numel
    # the gradient ranges still tile the bucket; the adapters' range comes first (the decoder's backward completes it)
    for chunks in (1, 4):
        rs = m.grad_ranges(chunks)
        assert rs[0] == (lp.base, m.proj.numel)
        cover = sorted(rs)
        assert cover[0][0] == 0 and cover[-1][1] == m.proj.numel and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    # peft's initialisation: B = 0 (the adapted model starts as the base model), A uniform within 1 / sqrt(in)
    sdl = lp.state_dict()
    assert key_of(0, "q_proj", "A") == "llm.base_model.model.model.layers.0.self_attn.q_proj.lora_A.default.weight"
    assert key_of(1, "down_proj", "B") == "llm.base_model.model.model.layers.1.mlp.down_proj.lora_B.default.weight"
    for k, v in sdl.items():
        if "lora_B" in k:
            assert float(v.abs().max()) == 0.0
        else:
            assert 0 < float(v.abs().max()) <= 1.0 / np.sqrt(v.shape[1]) + 1e-7
    assert lp.num_parameters() == sum(v.numel() for v in sdl.values())
    # layers are laid out in completion order: the last layer's tensors sit first
    assert lp.layer_range[geo.llm_layers - 1][0] == lp.base and lp.layer_range[0][1] == m.proj.numel


def test_lora_zero_b_equals_base_model_and_dropout_is_training_only():
    """With peft's init (B = 0) loss and logits are the un-adapted model's; dA is zero and dB is not.  In eval mode no mask
    is drawn (the dropout step counter does not move)."""
    geo = Geometry.from_dict(MID_GEOMETRY)
    sd = random_state_dict(geo, 2026, with_encoder=False)
    batch = synthetic_text_batch(geo, 2, seed=3, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=12, noise=False)
    base = TasuModel(geo, FakeOps(), "cpu")
    base.load_reference_state_dict(sd)
    sb = run_text(base, batch)
    m = TasuModel(geo, FakeOps(), "cpu")
    m.load_reference_state_dict(sd)
    m.enable_lora(LoraConfig(r=64, lora_alpha=16, lora_dropout=0.1))
    m.training = False
    sl = run_text(m, batch)
    assert int(m.lora.rng[1]) == 0
    assert torch.equal(sl.dev["loss_out"], sb.dev["loss_out"])
    assert torch.equal(m.logits_view(sl), base.logits_view(sb))
    for k, g in m.lora_grads().items():
        if "lora_A" in k:
            assert float(g.abs().max()) == 0.0, k
        else:
            assert float(g.abs().max()) > 0.0, k
    for k, g in m.projector_grads().items():
        assert cosine(g, base.projector_grads()[k]) > 0.9999, k
    m.training = True
    run_text(m, batch)
    assert int(m.lora.rng[1]) == 1
