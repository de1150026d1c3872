"""Audio front end, CPU side: the numpy oracle against an independent Kaldi-compatible implementation (golden generated
with transformers.audio_utils by oracle/make_golden_fbank.py), LFR / CMVN definitions, am.mvn parsing, and the host class
``ps_slm_amd.frontend.WavFrontend`` driven through the torch test double."""
import os

import numpy as np
import pytest
import torch

from fake_ops import FakeOps
from oracle import fbank_oracle as fo
from ps_slm_amd.frontend import WavFrontend, load_cmvn, mel_matrix

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "fbank_kaldi_hf.npz")


def test_oracle_fbank_matches_independent_kaldi_implementation():
    g = np.load(GOLDEN)
    got = fo.fbank(g["wave"])
    assert got.shape == g["fbank"].shape == (73, 80)
    # both sides are fp32/fp64 mixtures of the same formula; log-mel values are O(10), differences O(1e-5)
    assert np.abs(got - g["fbank"]).max() < 2e-4
    assert np.abs(fo.mel_banks() - g["mel"]).max() < 1e-6
    assert np.abs(mel_matrix(80, 512, 16000.0) - g["mel"]).max() < 1e-6


def test_lfr_definition():
    f = np.arange(20, dtype=np.float32)[:, None] * np.ones((1, 3), np.float32)
    out = fo.apply_lfr(f, 7, 6)
    assert out.shape == (4, 21)
    assert out[0, ::3].tolist() == [0, 0, 0, 0, 1, 2, 3]            # three copies of frame 0 on the left
    assert out[1, ::3].tolist() == [3, 4, 5, 6, 7, 8, 9]
    assert out[3, ::3].tolist() == [15, 16, 17, 18, 19, 19, 19]     # tail repeats the last frame
    assert fo.apply_lfr(f[:1], 7, 6).shape == (1, 21) and fo.apply_lfr(f[:0], 7, 6).shape == (0, 21)


def test_am_mvn_parse(tmp_path):
    means, scales = np.linspace(-9, -7, 14).astype(np.float32), np.linspace(0.1, 0.3, 14).astype(np.float32)
    vec = lambda v: "[ " + " ".join(f"{x:.6f}" for x in v) + " ]"
    p = tmp_path / "am.mvn"
    p.write_text("<Nnet>\n<Splice> 14 14\n[ 0 ]\n<AddShift> 14 14\n<LearnRateCoef> 0 " + vec(means) +
                 "\n<Rescale> 14 14\n<LearnRateCoef> 0 " + vec(scales) + "\n</Nnet>\n")
    m, s = load_cmvn(str(p))
    assert np.allclose(m, means, atol=1e-6) and np.allclose(s, scales, atol=1e-6)
    (tmp_path / "bad.mvn").write_text("<Nnet>\n</Nnet>\n")
    with pytest.raises(ValueError):
        load_cmvn(str(tmp_path / "bad.mvn"))


def test_wavfrontend_host_class_matches_oracle(tmp_path):
    g = np.load(GOLDEN)
    rng = np.random.default_rng(3)
    means, scales = rng.standard_normal(560).astype(np.float32), (rng.random(560) + 0.5).astype(np.float32)
    fe = WavFrontend(cmvn=(means, scales), ops=FakeOps(), device="cpu")
    for n in (399, 400, 559, 560, 12000):
        out, T = fe(g["wave"][:n])
        ref = fo.frontend(g["wave"][:n], means, scales)
        assert T == ref.shape[0] == fe.output_length(n) and tuple(out.shape) == ref.shape
        if T:
            assert np.abs(out.numpy() - ref).max() < 5e-4
    # config.yaml + am.mvn discovery
    (tmp_path / "config.yaml").write_text("frontend_conf:\n  fs: 16000\n  n_mels: 80\n  lfr_m: 5\n  lfr_n: 4\n  cmvn_file: /elsewhere/am.mvn\n")
    vec = lambda v: "[ " + " ".join(f"{x:.6f}" for x in v) + " ]"
    (tmp_path / "am.mvn").write_text("<AddShift> 400 400\n<LearnRateCoef> 0 " + vec(means[:400]) + "\n<Rescale> 400 400\n<LearnRateCoef> 0 " +
                                     vec(scales[:400]) + "\n")
    fe2 = WavFrontend.from_encoder_path(str(tmp_path), ops=FakeOps(), device="cpu")
    assert (fe2.lfr_m, fe2.lfr_n) == (5, 4) and fe2._means.numel() == 400
    out, T = fe2(g["wave"])
    assert tuple(out.shape) == (T, 400) and T == -(-73 // 4)
    assert np.abs(out.numpy() - fo.frontend(g["wave"], means[:400], scales[:400], lfr_m=5, lfr_n=4)).max() < 5e-4
    assert WavFrontend.from_encoder_path(None, device="cpu")._means is None          # no directory: defaults, no CMVN
