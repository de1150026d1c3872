"""Host logic on CPU (FakeOps double): plugin surface, config overrides, engine semantics (AdamW + DeepSpeed-style
WarmupCosineLR ordering), CPS noise RNG parity with the reference, and the N > 1 data-parallel path over gloo."""
import dataclasses
import math
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden_batch, load_npz, free_port
from fake_ops import FakeOps
from oracle import tasu_oracle as O
from ps_slm_amd.config import DEFAULT_DS_CONFIG, ModelConfig, RunConfig, TrainConfig, apply_overrides, load_ds_config
from ps_slm_amd.engine import TasuEngine, warmup_cosine_ratio
from ps_slm_amd.ps_slm import model_factory
from ps_slm_amd.synthetic import synthetic_text_batch


def make(seed=1234, noise=False, lr=None):
    tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, gt_emb_noise=noise, ctc_posterior=True, do_psd=True)
    mc = ModelConfig(llm_path="synthetic:mid", encoder_projector="linear-silu", llm_dim=256)
    model, tok = model_factory(tc, mc, device="cpu", ops=FakeOps(), init_seed=seed)
    cfg = load_ds_config(DEFAULT_DS_CONFIG)
    if lr is not None:
        cfg["lr"] = lr
    return model, tok, TasuEngine(model, cfg)


def to_call(raw):
    return dict(input_ids=raw["input_ids"], attention_mask=raw["attention_mask"], labels=raw["labels"],
                input_features=raw["input_features"], input_feature_length=raw["input_feature_length"],
                GT=[" ".join(map(str, p)) for p in raw["post_ids"]])


def test_config_overrides_match_script_syntax():
    cfg = apply_overrides(RunConfig(), ["++model_config.encoder_dim=25055", "++train_config.freeze_llm=true",
                                        "++train_config.gt_emb=true", "++train_config.use_fp16=false", "++metric=acc",
                                        "++dataset_config.train_max_frame_length=3000", "--local_rank=3",
                                        "hydra.run.dir=/tmp/x", "++model_config.llm_path=synthetic:mid",
                                        "++train_config.num_epochs=5", "++model_config.ctc_linear=null"])
    assert cfg.model_config.encoder_dim == 25055 and cfg.train_config.freeze_llm is True
    assert cfg.train_config.gt_emb is True and cfg.train_config.use_fp16 is False and cfg.train_config.num_epochs == 5
    assert cfg.dataset_config.train_max_frame_length == 3000 and cfg.model_config.ctc_linear is None
    assert cfg.train_config.get("ctc_posterior") is False and cfg.train_config.get("nope", 7) == 7
    with pytest.raises(ValueError):
        apply_overrides(RunConfig(), ["garbage"])


def test_factory_surface_and_errors():
    model, tok, eng = make()
    assert tok.pad_token_id == tok.eos_token_id and tok.default_ignore_token == -100
    assert sorted(model.state_dict()) == sorted("encoder_projector." + n for n in
                                                ("norm.weight", "norm.bias", "ffn.0.weight", "ffn.0.bias", "ffn.2.weight", "ffn.2.bias"))
    with pytest.raises(NotImplementedError):
        model_factory(TrainConfig(freeze_llm=True, gt_emb=True, ctc_posterior=True), ModelConfig(llm_path="synthetic:mid", encoder_projector="q-former", llm_dim=256),
                      device="cpu", ops=FakeOps())
    with pytest.raises(FileNotFoundError):
        model_factory(TrainConfig(freeze_llm=True, gt_emb=True, ctc_posterior=True), ModelConfig(llm_path="/nonexistent", encoder_projector="linear-silu"),
                      device="cpu", ops=FakeOps())
    # invalid attention mask (zeros on both sides) -> the reference's ValueError (ps-slm.py:785)
    raw = synthetic_text_batch(model.core.geo, 2, seed=1, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=8, noise=False)
    call = to_call(raw)
    am = call["attention_mask"].clone()
    am[0, 0] = False
    am[1, -1] = False
    call["attention_mask"] = am
    with pytest.raises(ValueError):
        model(**call)


def test_reference_loop_body_trains_the_plugin_through_autograd():
    """SURVEY 8b "Autograd glue" / VERDICT r5 item 6.  The reference's own loop (Multitask/finetune_deepspeed.py:127-149,
    Multitask/utils/deepspeed_utils.py:205-236) filters ``model.parameters()`` by ``requires_grad``, hands them to an optimizer,
    calls ``model(**batch)``, divides the loss and calls ``backward`` on it.  Here that works as it stands: ``outputs.loss`` is the
    result of a torch.autograd.Function whose backward runs the hand-scheduled HIP backward once and hands each leaf its slice of
    the flat gradient bucket.  ``torch.optim.AdamW`` over ``model.parameters()`` + ``loss.backward()`` reproduces TasuEngine's
    update (same lr schedule, betas, eps, decay) step for step."""
    from ps_slm_amd.ps_slm import EngineLoss
    model, _, _ = make(seed=77)
    ref_model, _, eng = make(seed=77)
    named = dict(model.named_parameters())
    assert sorted(named) == sorted(model.state_dict())
    params = list(filter(lambda p: p.requires_grad, model.parameters()))
    assert len(params) == 6 and all(p.is_leaf for p in params)
    assert all(a is b for a, b in zip(params, model.parameters()))              # the SAME leaves on every call
    assert sum(p.numel() for p in params) == model.core.proj.num_parameters()
    for k, v in model.state_dict().items():                  # views of the master buffer, reference shapes
        assert named[k].shape == v.shape and torch.equal(named[k].detach(), v)
    c = eng.cfg
    opt = torch.optim.AdamW(params, lr=c["lr"], betas=tuple(c["betas"]), eps=c["eps"], weight_decay=c["weight_decay"])
    geo = model.core.geo
    eng.sched_iter = eng.cfg["warmup_num_steps"] + 5         # past the warm-up (the schedule's very first ratio is 0)
    for step in range(4):
        raw = synthetic_text_batch(geo, 2, seed=30 + step, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=8, noise=False)
        # the engine of this path
        out_e, _ = eng(**to_call(raw))
        lr = eng.get_lr()[0]                                 # the rate eng.step() is about to use
        eng.backward(out_e.loss)
        eng.step()
        # the reference's loop body on the twin
        for g in opt.param_groups:
            g["lr"] = lr
        out, acc = model(**to_call(raw))
        assert out.loss.requires_grad and out.loss.grad_fn is not None and not isinstance(out.loss, EngineLoss)
        assert float(out.loss.detach()) == pytest.approx(float(out_e.loss.detach()), rel=1e-5)
        loss = out.loss / 2                                  # deepspeed_utils.py:210 (gradient_accumulation_steps = 2 there)
        opt.zero_grad()
        (loss * 2).backward()                                # ... whose factor the engine's 1/k would undo
        assert all(p.grad is not None and p.grad.shape == p.shape for p in params)
        for (n, p), (_, gv) in zip(model.named_parameters(), model._trainable_views(model.core.proj.g)):
            assert torch.allclose(p.grad, gv, rtol=1e-6, atol=0), n        # .grad = the bucket's slice x the incoming gradient
        opt.step()
        sd, sd_e = model.state_dict(), ref_model.state_dict()
        for k in sd:
            assert torch.allclose(sd[k], sd_e[k], rtol=2e-5, atol=2e-7), (step, k)
    # the optimizer wrote the masters behind the model's back: the next forward refreshed the bf16 working copies (the losses above
    # agree step after step only if it did), and the masters are what state_dict() exports
    # a second backward of the same step, or the backward of an older step, says what is wrong
    out, _ = model(**to_call(raw))
    out.loss.backward()
    from ps_slm_amd.ps_slm import _HipStep
    st = model.last_state
    with pytest.raises(RuntimeError, match="already run"):
        _HipStep.apply(st.dev["loss_out"][0], model, st, *params).backward()
    old, _ = model(**to_call(raw))
    model(**to_call(raw))
    with pytest.raises(RuntimeError, match="OLDER step"):
        old.loss.backward()
    # the engine's fast path and the autograd path do not mix on one step
    out3, _ = eng(**to_call(raw))
    eng.backward(out3.loss)
    with pytest.raises(RuntimeError, match="already run"):
        out3.loss.backward()
    eng.step()


def test_eval_mode_loss_and_generate_knobs_fail_loudly():
    """An eval-mode forward keeps no activations: its loss is an ``EngineLoss`` whose backward raises and says why (never a
    silent no-op); generate() rejects sampling / penalty knobs other than the reference's defaults instead of ignoring them."""
    from ps_slm_amd.ps_slm import EngineLoss
    model, _, eng = make()
    raw = synthetic_text_batch(model.core.geo, 2, seed=3, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=8, noise=False)
    model.eval()
    out, acc = model(**to_call(raw))
    loss = out.loss / 2
    assert isinstance(out.loss, EngineLoss) and isinstance(loss, EngineLoss) and loss.dim() == 0
    assert math.isfinite(float(loss.detach().float())) and not loss.requires_grad
    with pytest.raises(RuntimeError, match="eval mode"):
        loss.backward()
    model.train()
    for kw in (dict(do_sample=True), dict(top_p=0.9), dict(repetition_penalty=1.2), dict(temperature=0.7)):
        with pytest.raises(NotImplementedError, match=next(iter(kw))):
            model.generate(**to_call(raw), targets=["a"] * 2, **kw)


def test_noise_draws_replay_reference_rng():
    """Same seed -> the same (alpha, keep) as the reference drew in tests/golden/text_noise_right.npz."""
    b, z = golden_batch("text_noise_right")
    model, _, _ = make()
    # find the seed the generator settled on by replaying candidates exactly like oracle/make_golden.py
    for seed in range(777, 900):
        torch.manual_seed(seed)
        alphas, keeps = model.draw_noise(b["post_ids"])
        if np.allclose(alphas, z["alphas"], rtol=0, atol=0):
            assert np.array_equal(np.concatenate(keeps), z["keeps_flat"])
            return
    pytest.fail("no seed reproduced the reference's draws")


def test_engine_steps_match_oracle_adamw_and_schedule():
    model, _, eng = make(lr=2e-3)
    eng.cfg["warmup_num_steps"] = 4
    core = model.core
    raw = synthetic_text_batch(core.geo, 2, seed=4, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=8, noise=False, ragged=True)
    from ps_slm_amd.synthetic import random_state_dict  # oracle side: same weights through the state-dict export
    sd = {}
    # rebuild a reference-named state dict from the model's own weights for the oracle
    sd.update({k: v.clone() for k, v in model.state_dict().items()})
    llm = core.llm
    geo = core.geo
    H, G = geo.llm_heads, geo.llm_kv_heads
    sd["llm.model.embed_tokens.weight"] = llm.embed.clone()
    sd["llm.model.norm.weight"] = llm.norm.clone()
    for l, w in enumerate(llm.layers):
        p = f"llm.model.layers.{l}."
        q, k, v = w["wqkv"].float().split([H * 128, G * 128, G * 128], 0)
        bq, bk, bv = w["bqkv"].float().split([H * 128, G * 128, G * 128], 0)
        g_, u_ = w["wgu"].float().split([geo.llm_inter, geo.llm_inter], 0)
        sd.update({p + "input_layernorm.weight": w["ln1"], p + "post_attention_layernorm.weight": w["ln2"],
                   p + "self_attn.q_proj.weight": q, p + "self_attn.q_proj.bias": bq, p + "self_attn.k_proj.weight": k,
                   p + "self_attn.k_proj.bias": bk, p + "self_attn.v_proj.weight": v, p + "self_attn.v_proj.bias": bv,
                   p + "self_attn.o_proj.weight": w["wo"].float(), p + "mlp.gate_proj.weight": g_, p + "mlp.up_proj.weight": u_,
                   p + "mlp.down_proj.weight": w["wd"].float()})
    gd = dataclasses.asdict(geo)
    init = {k: sd[k].clone() for k in O.PROJ_KEYS}
    m = {k: torch.zeros_like(sd[k]) for k in O.PROJ_KEYS}
    v = {k: torch.zeros_like(sd[k]) for k in O.PROJ_KEYS}
    losses = []
    for step in range(1, 7):
        out, acc = eng(**to_call(raw))
        eng.backward(out.loss)
        eng.step()
        losses.append(float(out.loss.detach()))
        o_out, grads = O.loss_and_projector_grads(sd, raw, gd, "bf16")
        assert abs(losses[-1] - float(o_out["loss"])) < 1e-2, step
        lr = O.lr_for_optimizer_step(step, 2e-3, warmup_num_steps=4)
        assert abs(lr - eng.cfg["lr"] * warmup_cosine_ratio(step - 2, 4)) < 1e-12
        for k in O.PROJ_KEYS:
            O.adamw_step(sd[k], grads[k], m[k], v[k], step, lr)
    assert losses[1] == pytest.approx(losses[0], abs=1e-6), "DeepSpeed ordering: the first two optimizer steps run at lr = 0"
    assert losses[-1] < losses[0] - 1e-3
    mine = model.state_dict()
    # Adam normalises every element to ~lr, so bf16-level gradient noise moves near-zero-gradient elements freely:
    # compare the UPDATE (param - init) as a whole, not element by element.
    for k in O.PROJ_KEYS:
        da, db = (mine[k] - init[k]).flatten(), (sd[k] - init[k]).flatten()
        assert float(torch.nn.functional.cosine_similarity(da, db, dim=0)) > 0.98, k
        assert float(da.norm() / db.norm()) == pytest.approx(1.0, abs=0.05), k


def test_gradient_accumulation_matches_deepspeed_semantics():
    """gradient_accumulation_steps = 2: step() is a no-op off the boundary; on it the update equals ONE AdamW step on
    sum(g_i) / k^2 -- the reference's loop divides the loss by k (deepspeed_utils.py:210) and DeepSpeed's backward scales by
    1/k again -- and the scheduler advances once."""
    def build(ga):
        tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, gt_emb_noise=False, ctc_posterior=True, do_psd=True)
        mc = ModelConfig(llm_path="synthetic:mid", encoder_projector="linear-silu", llm_dim=256)
        model, _ = model_factory(tc, mc, device="cpu", ops=FakeOps(), init_seed=1234)
        cfg = load_ds_config(DEFAULT_DS_CONFIG)
        cfg.update(lr=1e-3, gradient_accumulation_steps=ga)
        eng = TasuEngine(model, cfg)
        eng.sched_iter = 10                  # past the zero-lr warm-up steps
        return model, eng
    geo = build(1)[0].core.geo
    raws = [synthetic_text_batch(geo, 2, seed=40 + i, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=8,
                                 noise=False) for i in range(2)]
    m2, e2 = build(2)
    p0 = m2.core.proj.p.clone()
    grads = []
    for i, raw in enumerate(raws):
        out, _ = e2(**to_call(raw))
        e2.backward(out.loss)
        grads.append(m2.core.proj.g.clone())
        e2.step()
        if i == 0:
            assert torch.equal(m2.core.proj.p, p0) and e2.global_steps == 0 and not e2.is_gradient_accumulation_boundary()
    assert e2.global_steps == 1 and e2.sched_iter == 11
    # reference update: one engine step with the averaged gradient planted in the bucket
    m1, e1 = build(1)
    out, _ = e1(**to_call(raws[0]))
    e1.backward(out.loss)
    m1.core.proj.g.copy_(0.25 * (grads[0] + grads[1]))
    e1.step()
    torch.testing.assert_close(m2.core.proj.p, m1.core.proj.p, rtol=0, atol=1e-7)
    assert float((m2.core.proj.p - p0).abs().max()) > 0


def _dp_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    model, _, eng = make(lr=1e-3)
    eng.sched_iter = 10                      # skip the zero-lr warm-up steps
    raw = synthetic_text_batch(model.core.geo, 2, seed=100 + rank, prompt_len=9, n_audio=21, target_len=17, speech_pos=4,
                               feat_frames=8, noise=False)
    # the rank's own gradient: one backward with the exchange switched off (the asynchronous all-reduces of the real
    # backward below rewrite the bucket in place, range by range)
    out, _ = eng(**to_call(raw))
    eng.exchange = False
    eng.backward(out.loss)
    g_local = model.core.proj.g.clone()
    eng.exchange, eng.micro_steps = True, 0
    out, _ = eng(**to_call(raw))
    eng.backward(out.loss)
    assert len(eng._pending) == eng.w1_chunks + 2 and eng.w1_chunks == 4      # tail, 4 row blocks of dW1, LayerNorm params
    assert sorted((lo, hi) for lo, hi, _ in eng._pending)[0][0] == 0
    assert sum(hi - lo for lo, hi, _ in eng._pending) == model.core.proj.numel   # the ranges tile the bucket
    eng.step()
    ret[rank] = dict(loss=float(out.loss.detach()), grad=g_local, param=model.core.proj.p.clone(),
                     reduced=eng.reduce_scalars(float(out.loss.detach())), joined=eng.all_have_data(rank == 0))
    dist.destroy_process_group()


def test_data_parallel_two_ranks_gloo():
    """DeepSpeed semantics: each rank's mean-CE gradient, AVERAGED over ranks, then one AdamW step; replicas stay equal."""
    world, port = 2, free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_worker, args=(world, port, ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    assert torch.equal(r0["param"], r1["param"]), "replicas diverged"
    assert r0["joined"] is False and r1["joined"] is False   # rank 1 had no data -> everybody stops (uneven-data join)
    assert r0["reduced"][0] == pytest.approx(r0["loss"] + r1["loss"], rel=1e-6)
    # single-process replay with the averaged gradient
    model, _, eng = make(lr=1e-3)
    eng.sched_iter = 10
    model.core.proj.g.copy_((r0["grad"] + r1["grad"]))
    eng.world = 2                              # grad_scale = 1/world inside the AdamW kernel
    eng._last_state = None
    eng.step()
    torch.testing.assert_close(model.core.proj.p, r0["param"], rtol=1e-6, atol=1e-7)


def test_plugin_loader_errors_match_reference():
    from ps_slm_amd.finetune_deepspeed import get_custom_model_factory
    with pytest.raises(ValueError):
        get_custom_model_factory(ModelConfig(file="model/ps-slm.txt:model_factory"))
    with pytest.raises(FileNotFoundError):
        get_custom_model_factory(ModelConfig(file="nope/ps_slm.py:model_factory"))
    with pytest.raises(AttributeError):
        get_custom_model_factory(ModelConfig(file="ps_slm_amd/ps_slm.py:not_there"))
    f = get_custom_model_factory(ModelConfig(file="ps_slm_amd/ps_slm.py:model_factory"))
    assert f.__name__ == "model_factory" and f.__module__ == "ps_slm.py"      # module name = file name, like the reference


_REFERENCE_LOADER = """
import importlib, importlib.machinery, importlib.util, sys
from pathlib import Path

def load_module_from_py_file(py_file):               # Multitask/utils/dataset_utils.py:14-25, restated
    module_name = Path(py_file).name
    loader = importlib.machinery.SourceFileLoader(module_name, py_file)
    spec = importlib.util.spec_from_loader(module_name, loader)
    module = importlib.util.module_from_spec(spec)
    loader.exec_module(module)
    return module

assert not any(p.rstrip("/").endswith("repo") for p in sys.path), sys.path
sys.path.insert(0, sys.argv[2])                      # tests/ only (fake_ops), NOT the repository root
model_mod = load_module_from_py_file(sys.argv[1] + "/ps_slm_amd/ps_slm.py")
data_mod = load_module_from_py_file(sys.argv[1] + "/ps_slm_amd/dataset.py")
factory = getattr(model_mod, "model_factory")        # Multitask/utils/model_utils.py:28-29
assert callable(getattr(data_mod, "get_speech_dataset"))
from fake_ops import FakeOps
from ps_slm_amd.config import ModelConfig, TrainConfig
tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, gt_emb_noise=False, ctc_posterior=True, do_psd=True)
mc = ModelConfig(llm_path="synthetic:mid", encoder_projector="linear-silu", llm_dim=256)
model, tok = factory(tc, mc, device="cpu", ops=FakeOps(), init_seed=1)
assert sorted(model.state_dict())[0].startswith("encoder_projector.")
print("LOADED", type(model).__name__, model_mod.__name__, data_mod.__name__)
"""


def test_reference_loader_loads_the_plugins(tmp_path):
    """The reference's own SourceFileLoader recipe (module name = file name, no package, repository root NOT on sys.path,
    another working directory) gets ``model_factory`` and ``get_speech_dataset`` out of the two plugin files and the
    factory builds a model."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "ref_loader.py"
    script.write_text(_REFERENCE_LOADER)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, str(script), repo, os.path.join(repo, "tests")], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "LOADED slam_model_asr ps_slm.py dataset.py" in r.stdout


def test_train_loop_and_checkpoint_roundtrip(tmp_path):
    """The entrypoint's loop body over the synthetic dataset (FakeOps double), then projector-only checkpoint
    save -> load into a fresh model through model_factory(ckpt_path=...)."""
    from ps_slm_amd.config import LogConfig
    from ps_slm_amd.finetune_deepspeed import SyntheticDataset, train
    model, tok, eng = make(lr=1e-3)
    eng.sched_iter = 10
    core = model.core
    ds = SyntheticDataset(core.geo, 2, 3, 0)
    import ps_slm_amd.synthetic as syn
    real = syn.synthetic_text_batch
    syn.synthetic_text_batch = lambda geo, B, seed, noise=False: real(geo, B, seed=seed, prompt_len=9, n_audio=21, target_len=17,
                                                                       speech_pos=4, feat_frames=8, noise=noise)
    try:
        # validation every 2 steps on a 2-batch eval split; improvement -> <output_dir>/<model_name>_epoch_1_step_2/
        tcfg = TrainConfig(num_epochs=1, run_validation=True, validation_interval=2, save_model=True,
                           output_dir=str(tmp_path / "out"), batching_strategy="dynamic")
        # use_wandb: the reference's wandb.log records (same keys / step numbering) in <wandb_dir>/metrics.jsonl (no wandb on the image)
        lcfg = LogConfig(log_interval=1, use_wandb=True, wandb_dir=str(tmp_path / "wandb"))
        res = train(eng, ds, tcfg, lcfg, 0, 1, eval_dataset=SyntheticDataset(core.geo, 2, 2, 0))
    finally:
        syn.synthetic_text_batch = real
    import json
    recs = [json.loads(l) for l in open(tmp_path / "wandb" / "metrics.jsonl")]
    inner = [r for r in recs if "train_inner/train_inner_loss" in r]
    assert [r["_step"] for r in inner] == [1, 2, 3] and all(0 <= r["train_inner/train_inner_accuracy"] <= 1 for r in inner)   # dynamic: step + 1
    valid = [r for r in recs if "valid/val_epoch_loss" in r]
    assert len(valid) == 1 and set(valid[0]) == {"valid/val_epoch_loss", "valid/val_perplexity", "valid/best_val_loss", "valid/val_accuracy",
                                                 "valid/val_best_accuracy"}
    assert valid[0]["valid/val_perplexity"] == pytest.approx(math.exp(valid[0]["valid/val_epoch_loss"]), rel=1e-6)
    ep = [r for r in recs if "train/train_epoch_loss" in r]
    assert len(ep) == 1 and ep[0]["train/train_epoch_loss"] == pytest.approx(sum(r["train_inner/train_inner_loss"] for r in inner) / 3, rel=1e-5)
    assert ep[0]["train/train_perplexity"] == pytest.approx(math.exp(ep[0]["train/train_epoch_loss"]), rel=1e-6)
    assert res["steps"] == 3 and res["avg_train_loss"] > 0
    assert res["avg_eval_loss"] > 0 and res["avg_eval_prep"] == pytest.approx(math.exp(res["avg_eval_loss"]), rel=1e-6)
    assert os.path.isfile(tmp_path / "out" / "asr_model_epoch_1_step_2" / "pytorch_model.bin")
    assert eng.module.training
    path = str(tmp_path / "pytorch_model.bin")
    eng.save_checkpoint(path)
    sd = torch.load(path)
    assert sorted(sd) == sorted(model.state_dict()) and sd["encoder_projector.ffn.0.weight"].shape == (128, 203)
    tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, ctc_posterior=True, do_psd=True)
    mc = ModelConfig(llm_path="synthetic:mid", encoder_projector="linear-silu", llm_dim=256)
    m2, _ = model_factory(tc, mc, device="cpu", ops=FakeOps(), init_seed=999, ckpt_path=path)
    for k, v in m2.state_dict().items():
        torch.testing.assert_close(v, sd[k])


def test_train_loop_announces_the_next_batch():
    """The loop holds one batch of lookahead and hands it to engine.prefetch() between forward and backward (on the GPU the audio
    branch then runs that batch's frozen encoder pass on a side stream, TasuModel.prefetch_encoder): every batch but the first is
    announced exactly once, right before it is consumed; the last step announces nothing.  On the CPU double prefetch() does
    nothing and returns False."""
    from ps_slm_amd.config import LogConfig
    from ps_slm_amd.finetune_deepspeed import SyntheticDataset, train
    model, tok, eng = make(lr=1e-3)
    ds = SyntheticDataset(model.core.geo, 2, 4, 0)
    import ps_slm_amd.synthetic as syn
    real = syn.synthetic_text_batch
    syn.synthetic_text_batch = lambda geo, B, seed, noise=False: real(geo, B, seed=seed, prompt_len=9, n_audio=21, target_len=17,
                                                                       speech_pos=4, feat_frames=8, noise=noise)
    events = []
    fwd, pre = eng.module.forward, eng.module.prefetch
    eng.module.__class__.__call__ = lambda self, **b: (events.append(("fwd", id(b["input_ids"]))), fwd(**b))[1]
    eng.module.prefetch = lambda **b: (events.append(("pre", id(b["input_ids"]))), pre(**b))[1]
    try:
        res = train(eng, ds, TrainConfig(num_epochs=1, batching_strategy="dynamic"), LogConfig(log_interval=10), 0, 1)
    finally:
        syn.synthetic_text_batch = real
        eng.module.__class__.__call__ = eng.module.__class__.forward
    assert res["steps"] == 4
    kinds = [k for k, _ in events]
    assert kinds == ["fwd", "pre", "fwd", "pre", "fwd", "pre", "fwd"]
    for n in range(1, len(events) - 1, 2):
        assert events[n][1] == events[n + 1][1]              # the announced batch is the very object the next forward gets
    assert eng.prefetch(**ds.collator(next(iter(ds)))) is False


def test_engine_refuses_to_train_less_than_asked():
    """freeze_encoder=false on the audio branch would, in the reference, leave the SenseVoice encoder trainable
    (ps-slm.py:31-40); this engine has no encoder backward and says so instead of silently training the projector only.  On the
    text-only branch (gt_emb=true) the encoder is not on the path and the flag is irrelevant."""
    mc = ModelConfig(llm_path="synthetic:mid", encoder_projector="linear-silu", llm_dim=256)
    tc = TrainConfig(freeze_llm=True, freeze_encoder=False, gt_emb=False, ctc_posterior=True, do_psd=True)
    model, _ = model_factory(tc, mc, device="cpu", ops=FakeOps(), init_seed=1)
    with pytest.raises(NotImplementedError, match="freeze_encoder"):
        TasuEngine(model, load_ds_config(DEFAULT_DS_CONFIG))
    tc2 = TrainConfig(freeze_llm=True, freeze_encoder=False, gt_emb=True, ctc_posterior=True, do_psd=True)
    model2, _ = model_factory(tc2, mc, device="cpu", ops=FakeOps(), init_seed=1)
    TasuEngine(model2, load_ds_config(DEFAULT_DS_CONFIG))


def test_batch_reader_thread_keeps_order_hands_over_failures_and_stops():
    """ps_slm_amd.finetune_deepspeed.BatchReader (one reader thread in place of the reference's DataLoader worker processes,
    Multitask/finetune_deepspeed.py:185-208): batches arrive collated and in dataset order, an exception in the reader surfaces in
    the consumer, and close() unblocks a producer that waits on the full queue."""
    import threading
    import time as _time

    from ps_slm_amd.finetune_deepspeed import BatchReader

    class DS:
        def __init__(self, n, fail_at=None):
            self.n, self.fail_at, self.reader_thread = n, fail_at, None

        def __iter__(self):
            self.reader_thread = threading.current_thread().name
            for i in range(self.n):
                if i == self.fail_at:
                    raise RuntimeError("corrupt utterance")
                yield [i, i + 100]

        def collator(self, raw):
            return {"input_ids": torch.tensor(raw), "input_features": None}

    ds = DS(20)
    got = [(raw, batch["input_ids"].tolist()) for raw, batch in BatchReader(ds, "cpu", depth=2)]
    assert got == [([i, i + 100], [i, i + 100]) for i in range(20)] and ds.reader_thread == "tasu-batch-reader"
    with pytest.raises(RuntimeError, match="corrupt utterance"):
        for _ in BatchReader(DS(10, fail_at=3), "cpu"):
            pass
    r = BatchReader(DS(10 ** 9), "cpu", depth=2)           # an endless source: the producer blocks on the full queue
    it = iter(r)
    next(it)
    t0 = _time.perf_counter()
    r.close()
    assert not r.thread.is_alive() and _time.perf_counter() - t0 < 5.0
