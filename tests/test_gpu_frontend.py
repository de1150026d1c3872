"""GPU parity of the audio front end (csrc/frontend.hip through the C-ABI) against oracle/fbank_oracle.py, and the chain
corpus -> ps_slm_amd.dataset -> WavFrontend (HIP) -> collator -> audio training step.

Tolerances.  fbank: the kernel runs an fp32 radix-2 FFT, the oracle an fp64 one; on log-mel values of O(10) the difference
is bounded by fp32 rounding of the power spectrum (relative 1e-5) except in bins whose energy is a cancellation residue,
hence abs 2e-3.  lfr_cmvn moves values and applies one add and one multiply in fp32: bit-exact."""
import json
import os
import random

import numpy as np
import pytest
import torch

import dataset_fixtures as fx
from fake_ops import FakeOps
from oracle import fbank_oracle as fo

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "fbank_kaldi_hf.npz")
FBANK_TOL = 2e-3


@pytest.fixture(scope="module")
def frontend():
    from ps_slm_amd.frontend import WavFrontend
    rng = np.random.default_rng(11)
    means, scales = rng.standard_normal(560).astype(np.float32), (rng.random(560) + 0.5).astype(np.float32)
    return WavFrontend(cmvn=(means, scales)), means, scales


def hip_fbank(fe, wave):
    w = torch.from_numpy(np.ascontiguousarray(wave, dtype=np.float32)).cuda()
    T = fe.num_frames(len(wave))
    fb = torch.full((max(T, 1), 80), float("nan"), device="cuda")
    fe(np.zeros(400, np.float32))                                            # constants onto the device
    fe.ops.fbank(w, len(wave), 32768.0, fe.win, fe.shift, fe._window, fe._mel, 80, 0.97, fb)
    torch.cuda.synchronize()
    return fb[:T].cpu().numpy()


@pytest.mark.parametrize("n", [400, 401, 559, 560, 5000, 12000])
def test_fbank_vs_oracle_and_independent_golden(frontend, n):
    fe = frontend[0]
    g = np.load(GOLDEN)
    got = hip_fbank(fe, g["wave"][:n])
    ref = fo.fbank(g["wave"][:n])
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert np.abs(got - ref).max() < FBANK_TOL
    if n == 12000:
        assert np.abs(got - g["fbank"]).max() < FBANK_TOL                    # transformers.audio_utils' Kaldi-compatible features


def test_fbank_silence_and_short_input(frontend):
    fe = frontend[0]
    got = hip_fbank(fe, np.zeros(2000, np.float32))
    floor = np.log(np.float32(1.1920928955078125e-07))                                             # the log floor
    assert np.abs(got - floor).max() <= 2e-6 and np.all(got == got[0, 0])                          # one ulp of device logf
    out, T = fe(np.zeros(399, np.float32))
    assert T == 0 and tuple(out.shape) == (0, 560)


def test_lfr_cmvn_bit_exact(frontend):
    fe, means, scales = frontend
    rng = np.random.default_rng(5)
    for T in (1, 5, 6, 7, 100, 2998):
        fb = rng.standard_normal((T, 80)).astype(np.float32) * 5 + 10
        for cm in (True, False):
            out = torch.empty(-(-T // 6), 560, device="cuda")
            fe.ops.lfr_cmvn(torch.from_numpy(fb).cuda(), T, 80, 7, 6, torch.from_numpy(means).cuda() if cm else None,
                            torch.from_numpy(scales).cuda() if cm else None, out)
            ref = fo.apply_lfr(fb, 7, 6)
            if cm:
                ref = fo.apply_cmvn(ref, means, scales)
            assert np.array_equal(out.cpu().numpy(), ref), (T, cm)


def test_full_size_30s_utterance(frontend):
    """BASELINE-size audio (30 s): parity against the oracle plus two size-independent properties -- frames are a pure
    function of their 400 samples (dropping k hops shifts the rows, bit-exactly) and gain g adds 2 ln g to every bin."""
    fe, means, scales = frontend
    n = 16000 * 30
    t = np.arange(n, dtype=np.float64) / 16000.0
    lcg = ((np.arange(n, dtype=np.int64) * 1103515245 + 12345) % 65536).astype(np.float64) / 32768.0 - 1.0
    wave = (0.2 * np.sin(2 * np.pi * (100.0 + 120.0 * t) * t) + 0.05 * lcg).astype(np.float32)
    out, T = fe(wave)
    assert T == 500 and tuple(out.shape) == (500, 560)
    assert np.abs(out.cpu().numpy() - fo.frontend(wave, means, scales)).max() < FBANK_TOL * float(scales.max())
    a, b = hip_fbank(fe, wave), hip_fbank(fe, wave[160 * 7:])
    assert a.shape[0] == 2998 and np.array_equal(a[7:], b)
    c = hip_fbank(fe, wave * np.float32(0.5))
    assert np.abs((a - c) - 2 * np.log(2.0)).max() < 1e-4                    # exact power of two gain: only log rounding differs


def test_dataset_to_training_step():
    """multitask.jsonl + wav-in-ark -> MultiTaskDataset with the HIP front end -> collator -> SANM encoder -> PSD ->
    projector -> LLM forward/backward, against the CPU double fed the oracle's features for the same corpus."""
    import tempfile
    from ps_slm_amd import dataset as ds_mod
    from ps_slm_amd.frontend import WavFrontend
    from ps_slm_amd.model import Geometry, TasuModel
    from ps_slm_amd.ops import HipOps
    from ps_slm_amd.synthetic import MID_GEOMETRY, random_state_dict
    from test_gpu_model import cosine, run_audio

    geo = Geometry.from_dict(dict(MID_GEOMETRY, feat_dim=560, speech_id=fx.SPEECH_ID, eos_id=fx.EOS_ID))
    sd = random_state_dict(geo, 77, with_encoder=True)
    rng = np.random.default_rng(2)
    cmvn = (-(rng.random(560).astype(np.float32) * 2 + 9), (rng.random(560).astype(np.float32) * 0.1 + 0.2))
    with tempfile.TemporaryDirectory() as root:
        dirs = fx.write_corpus(root)
        cfg = fx.dataset_config(root, dirs, False, 400)
        random.seed(0)
        ds = ds_mod.get_speech_dataset(cfg, fx.CharTokenizer(), "train", frontend=WavFrontend(cmvn=cmvn))
        raw = next(iter(ds))
        batch = ds.collator(raw)
        assert batch["input_features"].is_cuda and batch["input_features"].shape[0] == len(raw) > 1
        items = [json.loads(l) for l in open(os.path.join(dirs["train"], "multitask.jsonl"))][: len(raw)]
        feats = [fo.frontend(ds_mod.read_audio(it["path"])[1], *cmvn) for it in items]
    assert batch["input_feature_length"].tolist() == [f.shape[0] for f in feats]
    ref_feats = torch.zeros(batch["input_features"].shape)
    for i, f in enumerate(feats):
        ref_feats[i, : f.shape[0]] = torch.from_numpy(f)
    assert float((batch["input_features"].cpu() - ref_feats).abs().max()) < FBANK_TOL
    gm, cm = TasuModel(geo, HipOps(), "cuda"), TasuModel(geo, FakeOps(), "cpu")
    gm.load_reference_state_dict(sd)
    cm.load_reference_state_dict(sd)
    sg = run_audio(gm, batch)
    sc = run_audio(cm, dict(batch, input_features=ref_feats))
    assert np.array_equal(np.asarray(sg.dev["psd_lens"].cpu() if torch.is_tensor(sg.dev["psd_lens"]) else sg.dev["psd_lens"]),
                          np.asarray(sc.dev["psd_lens"]))
    assert abs(float(sg.dev["loss_out"][0]) - float(sc.dev["loss_out"][0])) < 5e-3
    gg, gc = gm.projector_grads(), cm.projector_grads()
    for k in gc:
        assert cosine(gg[k], gc[k]) > 0.999, k
