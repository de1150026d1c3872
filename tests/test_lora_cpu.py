"""LoRA recipe (use_peft=true), host logic on the CPU test double: the REAL host code of ps_slm_amd/lora.py + model.py driven
through tests/fake_ops.py, against goldens produced by the real reference model with the LoRA formula applied by hand
(oracle/make_golden_lora.py; peft is not available, see oracle/lora_oracle.py)."""
import numpy as np
import pytest
import torch

from conftest import free_port, load_npz
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
        if g.shape != ref.shape:                                  # the r = 64 fixtures keep every 2nd row / column
            g = g[::2, ::2]
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
        assert rs[0] == lp.layer_range[1] and rs[1] == lp.layer_range[0] and rs[0][0] == lp.base and rs[1][1] == m.proj.numel
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
    # B = 0 adds exact zeros to every accumulator; the K-extended GEMMs of the double sum in another blocking order than its
    # plain ones, so "equal" is up to fp32 summation order here (bitwise on the MFMA kernels: tests/test_gpu_lora.py)
    assert abs(float(sl.dev["loss_out"][0]) - float(sb.dev["loss_out"][0])) < 1e-4 and int(sl.dev["loss_out"][2]) == int(sb.dev["loss_out"][2])
    la, lb = m.logits_view(sl).float(), base.logits_view(sb).float()
    assert float((la != lb).float().mean()) < 2e-2 and float((la - lb).abs().max()) <= 2.0 ** -6 * float(lb.abs().max())
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


# ------------------------------------------------------------------------------------------------ plugin + engine surface
import os  # noqa: E402

import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from ps_slm_amd.config import DEFAULT_DS_CONFIG, ModelConfig, RunConfig, TrainConfig, apply_overrides, load_ds_config  # noqa: E402
from ps_slm_amd.engine import TasuEngine  # noqa: E402
from ps_slm_amd.ps_slm import model_factory  # noqa: E402


def make_lora(seed=1234, lr=1e-3, p=0.0, r=16, targets=None, **kw):
    tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, gt_emb_noise=False, ctc_posterior=True, do_psd=True, use_peft=True)
    tc.peft_config.r, tc.peft_config.lora_alpha, tc.peft_config.lora_dropout = r, 32, p
    if targets:
        tc.peft_config.target_modules = list(targets)
    mc = ModelConfig(llm_path="synthetic:mid", encoder_projector="linear-silu", llm_dim=256)
    model, tok = model_factory(tc, mc, device="cpu", ops=FakeOps(), init_seed=seed, **kw)
    cfg = load_ds_config(DEFAULT_DS_CONFIG)
    cfg["lr"] = lr
    eng = TasuEngine(model, cfg)
    eng.sched_iter = 10                      # past the zero-lr warm-up steps
    return model, eng


def to_call(raw):
    return dict(input_ids=raw["input_ids"], attention_mask=raw["attention_mask"], labels=raw["labels"],
                input_features=raw["input_features"], input_feature_length=raw["input_feature_length"],
                GT=[" ".join(map(str, p)) for p in raw["post_ids"]])


def test_peft_config_overrides_and_rejections():
    c = apply_overrides(RunConfig(), ["++train_config.use_peft=true", "++train_config.peft_config.r=32",
                                      "++train_config.peft_config.target_modules=[q_proj,v_proj]",
                                      "++train_config.peft_config.lora_dropout=0.1"])
    cfg = LoraConfig.from_peft_config(c.train_config.peft_config)
    assert (cfg.r, cfg.lora_alpha, cfg.lora_dropout, cfg.target_modules) == (32, 16.0, 0.1, ("q_proj", "v_proj"))
    assert cfg.scaling == 0.5
    # the reference's defaults (aispeech_asr_config.py:41-50)
    d = LoraConfig.from_peft_config(TrainConfig().peft_config)
    assert (d.r, d.lora_alpha, d.lora_dropout, len(d.target_modules)) == (64, 16.0, 0.05, 7)
    for bad in (dict(peft_method="prefix"), dict(bias="all"), dict(target_modules=["lm_head"]), dict(r=12),
                dict(inference_mode=True), dict(modules_to_save=["lm_head"]), dict(fan_in_fan_out=True), dict(task_type="SEQ_CLS")):
        with pytest.raises(NotImplementedError):           # peft's LoraConfig(**params) would act on every one of these keys
            LoraConfig.from_peft_config(bad)
    assert LoraConfig.from_peft_config(dict(task_type="CAUSAL_LM", inference_mode=False, r=8)).r == 8
    with pytest.raises(NotImplementedError, match="freeze_llm"):
        model_factory(TrainConfig(freeze_llm=False, gt_emb=True, ctc_posterior=True), ModelConfig(llm_path="synthetic:mid", encoder_projector="linear-silu", llm_dim=256),
                      device="cpu", ops=FakeOps())


def test_lora_engine_steps_and_checkpoint_roundtrip(tmp_path):
    model, eng = make_lora(p=0.1)
    core, lp = model.core, model.core.lora
    named = dict(model.named_parameters())
    assert len(named) == 6 + 2 * 7 * core.geo.llm_layers and all(p.requires_grad and p.is_leaf for p in named.values())
    assert sorted(named) == sorted(model.state_dict())
    assert sum(p.numel() for p in named.values()) == core.proj.num_parameters() + lp.num_parameters()
    raw = synthetic_text_batch(core.geo, 2, seed=5, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=8, noise=False)
    a0 = {k: v.clone() for k, v in lp.state_dict().items()}
    losses = []
    for step in range(6):
        out, _ = eng(**to_call(raw))
        losses.append(float(out.loss.detach()))
        eng.backward(out.loss)
        eng.step()
        now = lp.state_dict()
        if step == 0:
            # peft's init: B = 0, so the first step's dA is zero (A stays) while every B moves
            assert all(torch.equal(now[k], a0[k]) for k in now if "lora_A" in k)
            assert all(float(now[k].abs().max()) > 0 for k in now if "lora_B" in k)
        if step == 1:
            assert all(not torch.equal(now[k], a0[k]) for k in now if "lora_A" in k)
    assert losses[-1] < losses[0]
    assert int(lp.rng[1]) == 6                                # one mask draw per training forward
    # the bf16 working copies follow the master after every step
    assert torch.equal(core.proj.pb, core.proj.p.to(torch.bfloat16))
    t = "down_proj"
    assert torch.equal(lp.at[(1, t)][:, : lp.r], lp.view(core.proj.pb, 1, t, "A").t())
    sc = lp.cfg.scaling                                       # 32 / 16 = 2: folded into the copies (exact)
    assert float(lp.at[(1, t)][:, lp.r:].abs().max()) == 0.0
    assert torch.equal(lp.bts[(0, t)], (lp.view(core.proj.pb, 0, t, "B").t().float() * sc).to(torch.bfloat16))
    assert torch.equal(lp.as_[(0, t)], (lp.view(core.proj.pb, 0, t, "A").float() * sc).to(torch.bfloat16))
    # the K-extended weight of the down projection: [W | B | zero pad], K = 512 + 64 -> 640
    we = lp.wext[(1, "down")]
    I, D = core.geo.llm_inter, core.geo.llm_dim
    assert we.shape == (D, 640) and torch.equal(we[:, :I], core.llm.layers[1]["wd"])
    assert torch.equal(we[:, I:I + lp.r], lp.view(core.proj.pb, 1, t, "B")) and float(we[:, I + lp.r:].abs().max()) == 0.0
    wq = lp.wext[(0, "qkv")]                                  # q | k | v rows, B of each member in its own rank columns only
    H, G = core.geo.llm_heads, core.geo.llm_kv_heads
    assert torch.equal(wq[H * 128:(H + G) * 128, D + 64:D + 64 + lp.r], lp.view(core.proj.pb, 0, "k_proj", "B"))
    assert float(wq[H * 128:(H + G) * 128, D:D + 64].abs().max()) == 0.0 and float(wq[: H * 128, D + 64:].abs().max()) == 0.0
    # checkpoint: the trainable tensors under the reference's names; a fresh model built with ckpt_path continues from it
    path = str(tmp_path / "pytorch_model.bin")
    eng.save_checkpoint(path)
    sd = torch.load(path)
    assert sorted(sd) == sorted(model.state_dict())
    assert sd["llm.base_model.model.model.layers.1.mlp.gate_proj.lora_B.default.weight"].shape == (512, 16)
    m2, e2 = make_lora(seed=1234, p=0.1, ckpt_path=path)
    for k, v in m2.state_dict().items():
        assert torch.equal(v, sd[k]), k
    model.eval(), m2.eval()
    o1, _ = model(**to_call(raw))
    o2, _ = m2(**to_call(raw))
    assert float(o1.loss.detach()) == float(o2.loss.detach())
    # a projector-only checkpoint loads into the adapted model with the adapters reported missing (strict=False), and is refused strictly
    proj_only = {k: v for k, v in sd.items() if k.startswith("encoder_projector.")}
    missing, unexpected = m2.load_state_dict(proj_only)
    assert len(missing) == 2 * 7 * core.geo.llm_layers and not unexpected
    with pytest.raises(KeyError):
        m2.load_state_dict(proj_only, strict=True)


def _dp_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    model, eng = make_lora(targets=("q_proj", "v_proj", "down_proj"))
    sdl = random_lora_state_dict(model.core.geo, model.core.lora.cfg, 17)      # non-zero B: every adapter tensor has a gradient
    model.load_state_dict({**model.state_dict(), **sdl})
    raw = synthetic_text_batch(model.core.geo, 2, seed=100 + rank, prompt_len=9, n_audio=21, target_len=17, speech_pos=4,
                               feat_frames=8, noise=False)
    out, _ = eng(**to_call(raw))
    eng.exchange = False
    eng.backward(out.loss)
    g_local = model.core.proj.g.clone()
    eng.exchange, eng.micro_steps = True, 0
    out, _ = eng(**to_call(raw))
    eng.backward(out.loss)
    lp = model.core.lora
    assert len(eng._pending) == eng.w1_chunks + 4                 # the adapters of layer 1, of layer 0; tail, 4 row blocks of dW1, LayerNorm params
    assert (eng._pending[0][0], eng._pending[0][1]) == lp.layer_range[1] and (eng._pending[1][0], eng._pending[1][1]) == lp.layer_range[0]
    assert sum(hi - lo for lo, hi, _ in eng._pending) == model.core.proj.numel
    eng.step()
    ret[rank] = dict(grad=g_local, param=model.core.proj.p.clone())
    dist.destroy_process_group()


def test_lora_data_parallel_two_ranks_gloo():
    """The adapters' gradients travel in the same flat bucket: one range per span of decoder layers, issued as the backward
    completes them (before the projector's ranges); replicas stay equal and the update is AdamW on the rank-averaged gradient."""
    world, port = 2, free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_worker, args=(world, port, ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    assert torch.equal(r0["param"], r1["param"]), "replicas diverged"
    model, eng = make_lora(targets=("q_proj", "v_proj", "down_proj"))
    model.load_state_dict({**model.state_dict(), **random_lora_state_dict(model.core.geo, model.core.lora.cfg, 17)})
    lp = model.core.lora
    assert float(r0["grad"][lp.base:].abs().max()) > 0 and not torch.equal(r0["grad"][lp.base:], r1["grad"][lp.base:])
    model.core.proj.g.copy_(r0["grad"] + r1["grad"])
    eng.world = 2
    eng.step()
    torch.testing.assert_close(model.core.proj.p, r0["param"], rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------------ generate() with adapters
def gen_inputs():
    from conftest import split_flat
    zb, zl = load_npz("mid_generate_beam4"), load_npz("mid_generate_lora")
    geo = Geometry.from_dict(MID_GEOMETRY)
    cfg = LoraConfig(r=int(zl["r"]), lora_alpha=float(zl["alpha"]), lora_dropout=0.0, target_modules=tuple(str(zl["targets"]).split(",")))
    sd = random_state_dict(geo, int(zl["seed_w"]), with_encoder=False)
    lsd = random_lora_state_dict(geo, cfg, int(zl["seed_l"]))
    ids, am = torch.from_numpy(zb["input_ids"]), torch.from_numpy(zb["attention_mask"])
    return geo, cfg, sd, lsd, ids, am, split_flat(zb["post_ids_flat"], zb["post_lens"]), zl["tokens_text"], zb["tokens_text"]


def generate_text(model, ids, am, word_ids, **kw):
    from ps_slm_amd.decode import beam_search_generate
    st = model.prepare_text(ids, am, None, word_ids, None, None)
    model.forward_projector_text(st)
    return beam_search_generate(model, st, max_new_tokens=16, **kw).numpy()


def test_generate_with_adapters_vs_reference_tokens():
    """Beam-4 decode of the adapted model (prefill + loop on the merged weights W + s B A) against the tokens the reference's
    generate() produced with the LoRA formula applied by hand (fp32): a long common prefix (bf16 may flip a late near-tie), and
    NOT the un-adapted model's tokens.  With peft's zero-B init the merged weights are the base weights bit for bit."""
    geo, cfg, sd, lsd, ids, am, word_ids, ref, ref_base = gen_inputs()
    m = build(geo, cfg, sd, lsd, FakeOps(), "cpu")
    toks = generate_text(m, ids, am, word_ids)
    common = (toks == ref).cumprod(1).sum(1)
    assert (common >= 8).all(), (toks, ref)
    assert not np.array_equal(toks[:, :4], ref_base[:, :4])
    assert m._lora_run is not None and m.llm is not m.lora._merged            # the training-step weights are back in place
    # zero-B adapters: exactly the base model's decode
    base = TasuModel(geo, FakeOps(), "cpu")
    base.load_reference_state_dict(sd)
    z = TasuModel(geo, FakeOps(), "cpu")
    z.load_reference_state_dict(sd)
    z.enable_lora(cfg)
    assert np.array_equal(generate_text(z, ids, am, word_ids), generate_text(base, ids, am, word_ids))
    # the merged weights follow the adapters: after a load they are rebuilt (in place)
    before = z.lora._merged.layers[0]["wqkv"].clone()
    ptr = z.lora._merged.layers[0]["wqkv"].data_ptr()
    z.lora.load_state_dict(lsd)
    z.sync_projector_copies()
    toks2 = generate_text(z, ids, am, word_ids)
    assert z.lora._merged.layers[0]["wqkv"].data_ptr() == ptr and not torch.equal(z.lora._merged.layers[0]["wqkv"], before)
    assert np.array_equal(toks2, toks)


def test_lora_on_the_audio_path_and_with_other_projectors():
    """The adapters sit in the decoder, so every front end keeps working under use_peft: the audio branch (encoder -> PSD ->
    projector) and the alternate projectors; one engine step each moves projector and adapters and lowers the loss."""
    from conftest import mid_audio_psd_case
    geo, sd, batch, z = mid_audio_psd_case()
    cfg = LoraConfig(r=16, lora_alpha=32, lora_dropout=0.1)
    m = build(geo, cfg, sd, random_lora_state_dict(geo, cfg, 3), FakeOps(), "cpu")
    losses = []
    for step in range(1, 4):
        st = m.prepare_audio(batch["input_ids"], batch["attention_mask"], batch["labels"], batch["input_features"], batch["input_feature_length"])
        m.forward_llm(st)
        m.backward(st)
        assert np.array_equal(st.dev["psd_lens"], z["psd_lens"])
        losses.append(float(st.dev["loss_out"][0]))
        g = m.proj.g
        assert torch.isfinite(g).all() and float(g[m.lora.base:].abs().max()) > 0 and float(g[: m.lora.base].abs().max()) > 0
        m.ops.adamw(m.proj.p, m.proj.g, m.proj.m, m.proj.v, m.proj.pb, torch.tensor([2e-3]), 0.9, 0.999, 1e-6, 0.0, step, 1.0)
        m.refresh_working_copies()
    assert losses[-1] < losses[0]
    # the cross-attention projector (its keys / values are the frozen embedding table) next to adapters
    from conftest import ca_projector_case
    geo2, sd2, batch2, _ = ca_projector_case()
    cfg2 = LoraConfig(r=8, lora_alpha=8, lora_dropout=0.0, target_modules=("q_proj", "v_proj", "up_proj"))
    m2 = build(geo2, cfg2, sd2, random_lora_state_dict(geo2, cfg2, 4), FakeOps(), "cpu")
    st2 = run_text(m2, batch2)
    assert torch.isfinite(st2.dev["loss_out"]).all() and float(m2.proj.g[m2.lora.base:].abs().max()) > 0


def rank_gt64_case():
    geo = Geometry.from_dict(MID_GEOMETRY)
    cfg = LoraConfig(r=128, lora_alpha=64, lora_dropout=0.0, target_modules=("q_proj", "k_proj", "o_proj", "down_proj"))
    sd = random_state_dict(geo, 2026, with_encoder=False)
    lsd = random_lora_state_dict(geo, cfg, 12, b_scale=0.03)
    batch = synthetic_text_batch(geo, 2, seed=8, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=12, noise=False, ragged=True)
    return geo, cfg, sd, lsd, batch


def test_lora_rank_above_64_against_torch_autograd():
    """r = 128 (two 64-wide rank blocks per member; the rank GEMMs fall back to the tile policy above N = 64): the adapters'
    gradients of the double against torch autograd through the SAME decoder restated with the LoRA formula (oracle, fp32)."""
    import dataclasses
    from oracle import tasu_oracle as O
    geo, cfg, sd, lsd, batch = rank_gt64_case()
    m = build(geo, cfg, sd, lsd, FakeOps(), "cpu")
    st = run_text(m, batch)
    # oracle: fold the adapters into fp32 weights W + s B A and differentiate with respect to A and B
    W = {k: v.clone() for k, v in sd.items()}
    leaves = {k: v.clone().requires_grad_(True) for k, v in lsd.items()}
    from ps_slm_amd.lora import key_of
    for l in range(geo.llm_layers):
        for t in cfg.target_modules:
            parent = "self_attn" if t in ("q_proj", "k_proj", "v_proj", "o_proj") else "mlp"
            name = f"llm.model.layers.{l}.{parent}.{t}.weight"
            W[name] = sd[name] + cfg.scaling * leaves[key_of(l, t, "B")] @ leaves[key_of(l, t, "A")]
    out = O.forward_text(W, batch, dataclasses.asdict(geo), "fp32")
    grads = torch.autograd.grad(out["loss"], list(leaves.values()))
    assert abs(float(st.dev["loss_out"][0]) - float(out["loss"])) < 2e-2
    mine = m.lora_grads()
    for k, g in zip(leaves, grads):
        assert cosine(mine[k], g) > 0.99, k


def test_freeze_projector_trains_the_adapters_only(tmp_path):
    """train_config.freeze_projector=true (the shipped script's knob, honoured for linear-silu like ps-slm.py:47-54) next to
    use_peft: the projector's tensors carry requires_grad=False, get no weight gradient, no exchange range and no optimizer
    update, and the checkpoint holds the adapters only (what exclude_frozen_parameters keeps); a stage-1 projector checkpoint
    still loads.  Without use_peft nothing would be left to train: ValueError."""
    tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, gt_emb_noise=False, ctc_posterior=True, do_psd=True, use_peft=True,
                     freeze_projector=True)
    tc.peft_config.r, tc.peft_config.lora_dropout = 16, 0.0
    mc = ModelConfig(llm_path="synthetic:mid", encoder_projector="linear-silu", llm_dim=256)
    model, _ = model_factory(tc, mc, device="cpu", ops=FakeOps(), init_seed=1234)
    core, lp = model.core, model.core.lora
    cfg = load_ds_config(DEFAULT_DS_CONFIG)
    cfg["lr"] = 1e-3
    eng = TasuEngine(model, cfg)
    eng.sched_iter = 10
    named = dict(model.named_parameters())
    assert all(p.requires_grad == ("lora_" in k) for k, p in named.items())
    assert sorted(model.state_dict()) == sorted(k for k in named if "lora_" in k)
    assert [r[0] for r in core.grad_ranges(4)][0] == lp.base and sum(hi - lo for lo, hi in core.grad_ranges(4)) == lp.numel
    raw = synthetic_text_batch(core.geo, 2, seed=5, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=8, noise=False)
    p0 = core.proj.p.clone()
    for _ in range(3):
        out, _ = eng(**to_call(raw))
        eng.backward(out.loss)
        eng.step()
    assert torch.equal(core.proj.p[: lp.base], p0[: lp.base]) and not torch.equal(core.proj.p[lp.base:], p0[lp.base:])
    assert float(core.proj.g[: lp.base].abs().max()) == 0.0                      # no projector weight gradient was ever computed
    path = str(tmp_path / "adapters.bin")
    eng.save_checkpoint(path)
    assert all("lora_" in k for k in torch.load(path))
    # a stage-1 projector checkpoint loads into the frozen projector (strict=False, like ps-slm.py:163-170)
    stage1 = {"encoder_projector." + n: core.proj.export(n) + 0.5 for n in core.proj.names}
    missing, unexpected = model.load_state_dict(stage1)
    assert not unexpected and all("lora_" in k for k in missing)
    assert torch.equal(core.proj.export("ffn.2.bias"), stage1["encoder_projector.ffn.2.bias"])
    with pytest.raises(ValueError, match="nothing to train"):
        model_factory(TrainConfig(freeze_llm=True, gt_emb=True, ctc_posterior=True, freeze_projector=True), mc, device="cpu", ops=FakeOps())
    # the flag is the reference's for linear-silu only: another projector ignores it
    m2, _ = model_factory(TrainConfig(freeze_llm=True, gt_emb=True, ctc_posterior=True, freeze_projector=True),
                          ModelConfig(llm_path="synthetic:mid", encoder_projector="linear", encoder_projector_ds_rate=1, llm_dim=256), device="cpu", ops=FakeOps())
    assert m2.core.freeze_projector is False


def test_dropout_flag_survives_a_graph_replay_of_the_forward():
    """``st.lora_drop`` was set inside forward_llm, i.e. only when the Python body ran.  A hipGraph replay of the
    forward (no Python) left a fresh StepState at False, and a backward captured right then regenerated no dropout masks for that
    shape -- silently wrong adapter gradients from then on.  The flag is now recorded by run_forward_* outside the captured
    region: emulate a replay (``_graphed`` that does not call its function) and check the step state."""
    z, geo, cfg, sd, lsd, _ = golden_case("mid_text_lora_drop")
    model = build(geo, cfg, sd, lsd, FakeOps(), "cpu", rng=z["rng"])
    model.training = True
    calls = []
    model._graphed = lambda key, fn, st: calls.append(key)                 # a replay: the captured launches run, Python does not
    st = type("St", (), {"lora_drop": False, "dev": {}, "path": "text", "B": 1, "S": 1, "Ra": 0, "Rap": 0, "Fap": 0, "nLp": 0})()
    model.run_forward_text(st)
    assert calls and st.lora_drop is True
    model.training = False
    st2 = type("St", (), {"lora_drop": True, "dev": {}, "path": "text", "B": 1, "S": 1, "Ra": 0, "Rap": 0, "Fap": 0, "nLp": 0})()
    model.run_forward_llm(st2)
    assert st2.lora_drop is False                                          # eval mode: no masks, whatever the state held before


def decode_lora_margin(model, geo, cases):
    """[(case, got, want)] of the cases of tests/golden/mid_generate_lora_margin.npz whose tokens differ from the reference's."""
    from ps_slm_amd.decode import beam_search_generate
    bad = []
    for n, c in enumerate(cases):
        st = model.prepare_text(c["ids"], c["am"], None, c["post_ids"], None, None)
        model.forward_projector_text(st)
        toks = beam_search_generate(model, st, eos_token_id=geo.eos_id, pad_token_id=geo.eos_id, **c["kw"]).numpy()
        if toks.shape != c["tokens"].shape or not np.array_equal(toks, c["tokens"]):
            bad.append((n, toks.tolist(), c["tokens"].tolist()))
    return bad


def test_generate_with_adapters_margin_cases_exact_on_the_double():
    """the decode of the ADAPTED model (merged weights W' = bf16(W + s B A), ps_slm_amd/lora.py:merged_llm) must
    EQUAL the tokens of the reference's generate() with the LoRA formula applied by hand, on the 7 rounding-stable cases of
    oracle/make_golden_generate_lora_margin.py (1-4 beams, min_length, length penalties, a 40-position case).  Every case decodes
    to something else without the adapters, so equality says the adapters are in the decode."""
    from conftest import decode_lora_margin_cases
    geo, cfg, sd, lsd, cases = decode_lora_margin_cases()
    assert len(cases) == 7 and all(c["tokens"].shape != c["tokens_base"].shape or not np.array_equal(c["tokens"], c["tokens_base"])
                                   for c in cases)
    m = build(geo, cfg, sd, lsd, FakeOps(), "cpu")
    assert not decode_lora_margin(m, geo, cases)
    base = TasuModel(geo, FakeOps(), "cpu")
    base.load_reference_state_dict(sd)
    assert len(decode_lora_margin(base, geo, cases)) == len(cases)            # (and the fixture's tokens_base are the base model's)
    for c in cases:
        st = base.prepare_text(c["ids"], c["am"], None, c["post_ids"], None, None)
        base.forward_projector_text(st)
        from ps_slm_amd.decode import beam_search_generate
        tb = beam_search_generate(base, st, eos_token_id=geo.eos_id, pad_token_id=geo.eos_id, **c["kw"]).numpy()
        assert np.array_equal(tb, c["tokens_base"])
