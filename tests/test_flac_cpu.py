"""FLAC decoding for the dataset path (``.flac`` entries, speech_dataset_large.py:123-127 -> torchaudio.load + channel mean): the
native decoder of libtasu_hip.so (csrc/flac.hip, host code) against streams written by tests/flac_fixtures.py -- every subframe
type, residual coding, stereo mode -- plus corruption detection (frame CRC, STREAMINFO MD5).  Host-only: runs without a GPU."""
import numpy as np
import pytest

from flac_fixtures import write_flac
from ps_slm_amd.dataset import audio_num_samples, read_audio


def signal(T, C, bps, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(T)
    amp = (1 << (bps - 1)) * 0.4
    x = amp * np.sin(2 * np.pi * 220 * t / 16000)[:, None] * np.linspace(1, 0.3, C)[None, :]
    x = x + rng.normal(0, amp * 0.02, (T, C))
    x[T // 3: T // 3 + 1024] = 0                          # a silent stretch: CONSTANT subframes really occur
    x[2 * T // 3: 2 * T // 3 + 1024] = (x[2 * T // 3: 2 * T // 3 + 1024] // 4) * 4   # multiples of 4: wasted bits can occur
    return np.clip(np.round(x), -(1 << (bps - 1)), (1 << (bps - 1)) - 1).astype(np.int64)


@pytest.mark.parametrize("C,bps,blocksize,T", [(1, 16, 1024, 20000), (2, 16, 1024, 23456), (2, 16, 4096, 30000), (1, 16, 576, 7000),
                                                (2, 24, 1000, 9000), (1, 8, 256, 3000)])
def test_flac_round_trip(tmp_path, C, bps, blocksize, T):
    x = signal(T, C, bps, seed=T + C)
    p = tmp_path / "a.flac"
    p.write_bytes(write_flac(x if C > 1 else x[:, 0], rate=16000, bps=bps, blocksize=blocksize, seed=bps + C))
    rate, wav = read_audio(str(p))
    assert rate == 16000 and wav.dtype == np.float32 and len(wav) == T and audio_num_samples(str(p)) == T
    want = (x.astype(np.float32) / np.float32(1 << (bps - 1))).mean(1) if C > 1 else x[:, 0].astype(np.float32) / np.float32(1 << (bps - 1))
    np.testing.assert_allclose(wav, want, rtol=0, atol=2 ** -(bps - 1) * 1e-3 + 1e-7)


def test_flac_corruption_is_detected(tmp_path):
    x = signal(12000, 2, 16, seed=5)
    blob = bytearray(write_flac(x, blocksize=1024, seed=1))
    p = tmp_path / "bad.flac"
    bad = bytearray(blob)
    bad[len(bad) // 2] ^= 0x10                           # a flipped bit inside a frame: CRC-16 (or a parse error) must catch it
    p.write_bytes(bad)
    with pytest.raises(ValueError):
        read_audio(str(p))
    bad = bytearray(blob)
    bad[4 + 4 + 18 + 3] ^= 0xFF                          # STREAMINFO MD5 no longer matches the (correctly decoded) PCM
    p.write_bytes(bad)
    with pytest.raises(ValueError):
        read_audio(str(p))
    p.write_bytes(b"RIFFnot flac at all.........................................")
    with pytest.raises(ValueError):
        read_audio(str(p))
    # without an MD5 (all zero: "unknown") the stream still decodes
    p.write_bytes(write_flac(x, blocksize=1024, seed=1, with_md5=False))
    assert len(read_audio(str(p))[1]) == len(x)
