"""ps_slm_amd/dataset.py against golden batches produced by the REAL reference dataset
(oracle/make_golden_dataset.py -> tests/golden/dataset_*.npz) on the same deterministic corpus: sharding, prompt choice,
templating, target cleaning, labels, padding side, feature padding and frame-budget batching must match exactly."""
import json
import os
import random

import numpy as np
import pytest
import torch
import torch.distributed as dist

import dataset_fixtures as fx
from ps_slm_amd import dataset as ds_mod

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def run_case(tmp_path, name, text_only=False, monkeypatch=None):
    split, infer, budget, world, rank, seed = fx.CASES[name]
    dirs = fx.write_corpus(str(tmp_path))
    cfg = fx.dataset_config(str(tmp_path), dirs, infer, budget)
    cfg.text_only = text_only
    if world > 1:
        monkeypatch.setattr(dist, "is_initialized", lambda: True)
        monkeypatch.setattr(dist, "get_world_size", lambda *a: world)
        monkeypatch.setattr(dist, "get_rank", lambda *a: rank)
    random.seed(seed)
    ds = ds_mod.get_speech_dataset(cfg, fx.CharTokenizer(), split, frontend=fx.StandInFrontend())
    return ds, [ds.collator(raw) for raw in ds]


@pytest.mark.parametrize("name", list(fx.CASES))
def test_batches_match_reference(tmp_path, monkeypatch, name):
    ds, batches = run_case(tmp_path, name, monkeypatch=monkeypatch)
    gold = np.load(os.path.join(GOLDEN, f"dataset_{name}.npz"))
    assert int(gold["n_batches"]) == len(batches)
    assert int(gold["dataset_len"]) == len(ds)
    got = fx.flatten_batches(batches)
    assert sorted(got) == sorted(k for k in gold.files if k != "dataset_len")
    for k, v in got.items():
        g = gold[k]
        if v.dtype.kind in "US":
            assert json.loads(str(v)) == json.loads(str(g)), k
        else:
            assert v.dtype == g.dtype and v.shape == g.shape, (k, v.dtype, g.dtype, v.shape, g.shape)
            assert np.array_equal(v, g), k


def test_text_only_keeps_batch_composition(tmp_path):
    """text_only reads only the audio length; batches (ids, labels, lengths, grouping) are those of the full path."""
    _, full = run_case(tmp_path / "a", "train_w1")
    _, lite = run_case(tmp_path / "b", "train_w1", text_only=True)
    assert len(full) == len(lite)
    for f, l in zip(full, lite):
        assert l["input_features"] is None
        for k in ("input_ids", "attention_mask", "labels", "input_feature_length"):
            assert torch.equal(f[k], l[k])
        assert f["GT"] == l["GT"]


def test_read_audio_wav_and_ark(tmp_path):
    import wave
    dirs = fx.write_corpus(str(tmp_path))
    item = json.loads(open(os.path.join(dirs["train"], "multitask.jsonl")).readline())
    rate, x = ds_mod.read_audio(item["path"])
    assert rate == 16000 and np.array_equal((x * 32768).astype(np.int16), fx.waveform_i16(0))
    assert ds_mod.audio_num_samples(item["path"]) == fx.N_SAMPLES[0]
    p = str(tmp_path / "stereo.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(2), w.setsampwidth(2), w.setframerate(16000)
        w.writeframes(np.stack([fx.waveform_i16(1), fx.waveform_i16(1)], 1).tobytes())
    rate, y = ds_mod.read_audio(p)
    assert len(y) == fx.N_SAMPLES[1] and ds_mod.audio_num_samples(p) == fx.N_SAMPLES[1]
    # .flac entries go through the native decoder (tests/test_flac_cpu.py); here: same samples as the wav of the same waveform
    from flac_fixtures import write_flac
    pf = tmp_path / "u1.flac"
    pf.write_bytes(write_flac(fx.waveform_i16(1).astype(np.int64), rate=16000, bps=16, blocksize=1024))
    rate, z = ds_mod.read_audio(str(pf))
    assert rate == 16000 and np.array_equal(z, fx.waveform_i16(1).astype(np.float32) / 32768.0)
    assert ds_mod.audio_num_samples(str(pf)) == fx.N_SAMPLES[1]


def test_window_class_first_element_and_budget():
    e = lambda n, f: {"input_ids": torch.zeros(n), "input_feature_length": f}
    assert ds_mod.window_class(e(10, 80), [], 1000, 8) is True
    assert not ds_mod.window_class(e(10, 80), [e(10, 80)], 38, 8)          # 2 * (10 + 10 - 1) = 38 fits
    assert ds_mod.window_class(e(10, 80), [e(10, 80)], 37, 8)


def test_entrypoint_trains_from_jsonl_corpus_text_only(tmp_path):
    """finetune_deepspeed's loader + loop on a jsonl corpus through ``dataset_config.file=ps_slm_amd/dataset.py:...`` in
    text_only mode (audio lengths only): every dynamic batch goes through the text pseudo-posterior step (FakeOps double)."""
    from fake_ops import FakeOps
    from ps_slm_amd.config import DEFAULT_DS_CONFIG, LogConfig, ModelConfig, TrainConfig, load_ds_config
    from ps_slm_amd.engine import TasuEngine
    from ps_slm_amd.finetune_deepspeed import get_dataset, train
    from ps_slm_amd.ps_slm import model_factory

    class Tok(fx.CharTokenizer):
        eos_token_id, pad_token_id = 980, 981

        def encode(self, text):
            return [990 if t == fx.SPEECH_ID else t for t in super().encode(text)]

    tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, ctc_posterior=True, do_psd=True, num_epochs=1)
    mc = ModelConfig(llm_path="synthetic:mid", encoder_projector="linear-silu", llm_dim=256)
    model, _ = model_factory(tc, mc, device="cpu", ops=FakeOps(), init_seed=5)
    dirs = fx.write_corpus(str(tmp_path), split_sizes=(("train", 4),))
    cfg = fx.dataset_config(str(tmp_path), dirs, False, 400)
    cfg.text_only, cfg.file = True, "ps_slm_amd/dataset.py:get_speech_dataset"
    random.seed(1)
    ds = get_dataset(cfg, Tok(), "train", model.core.geo, 0)
    assert type(ds).__module__ == "dataset.py"             # loaded like the reference does: module name = file name
    n_batches = sum(1 for _ in ds)
    res = train(TasuEngine(model, load_ds_config(DEFAULT_DS_CONFIG)), ds, tc, LogConfig(log_interval=1), 0, 1)
    assert res["steps"] == n_batches >= 2 and np.isfinite(res["avg_train_loss"]) and res["avg_train_loss"] > 0
    # audio branch + text_only is a configuration error, not a silent skip
    tc2 = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=False, ctc_posterior=True, do_psd=True)
    m2, _ = model_factory(tc2, mc, device="cpu", ops=FakeOps(), init_seed=5, with_encoder=True)
    with pytest.raises(ValueError, match="text_only"):
        m2(**ds.collator(next(iter(ds))))
