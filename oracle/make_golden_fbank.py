"""TEST INFRASTRUCTURE ONLY -- tests/golden/fbank_kaldi_hf.npz: Kaldi-compatible log-mel features of a short deterministic
waveform computed by an INDEPENDENT implementation, ``transformers.audio_utils`` (mel_filter_bank(mel_scale="kaldi",
triangularize_in_mel_space=True) + spectrogram(preemphasis, remove_dc_offset, mel_floor) -- the numpy path that
Speech2TextFeatureExtractor uses in place of torchaudio.compliance.kaldi.fbank and that is tested upstream against it).
funasr and torchaudio themselves are not installed, so this is the closest available pin for oracle/fbank_oracle.py::fbank;
LFR and CMVN have no independent implementation here and stay restated-only.

Run in the build container only:  python oracle/make_golden_fbank.py   (transformers 4.x installed there)
"""
import os

import numpy as np
from transformers.audio_utils import mel_filter_bank, spectrogram, window_function

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wave(n=12000):
    t = np.arange(n, dtype=np.float64) / 16000.0
    chirp = 0.3 * np.sin(2 * np.pi * (200.0 + 3000.0 * t) * t)
    buzz = 0.05 * np.sign(np.sin(2 * np.pi * 110.0 * t))
    lcg = (np.arange(n, dtype=np.int64) * 1103515245 + 12345) % 65536
    noise = 0.02 * (lcg.astype(np.float64) / 32768.0 - 1.0)
    env = np.where((t > 0.30) & (t < 0.36), 0.0, 1.0)                      # a silent gap: exercises the log floor region
    return ((chirp + buzz + noise) * env + 0.01).astype(np.float32)         # + DC offset


def main():
    x = test_wave()
    mel = mel_filter_bank(num_frequency_bins=257, num_mel_filters=80, min_frequency=20, max_frequency=8000,
                          sampling_rate=16000, norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
    feats = spectrogram(x * 32768.0, window_function(400, "hamming", periodic=False), frame_length=400, hop_length=160,
                        fft_length=512, power=2.0, center=False, preemphasis=0.97, mel_filters=mel, log_mel="log",
                        mel_floor=1.192092955078125e-07, remove_dc_offset=True).T
    path = os.path.join(ROOT, "tests", "golden", "fbank_kaldi_hf.npz")
    np.savez_compressed(path, wave=x, fbank=feats.astype(np.float32), mel=mel.T.astype(np.float32))
    print(feats.shape, f"{os.path.getsize(path) / 1024:.1f} KB")


if __name__ == "__main__":
    main()
