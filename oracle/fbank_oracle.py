"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the audio front end the reference runs inside its dataset
(Multitask/dataset/speech_dataset_large.py:133-146): funasr ``WavFrontend`` = Kaldi-compatible log-mel filterbank
(torchaudio.compliance.kaldi.fbank), low-frame-rate stacking (``apply_lfr``) and CMVN (``apply_cmvn``).

PARITY UNPINNED: funasr and torchaudio are third-party dependencies that are neither in the reference tree nor installed
here, and the reference holds no feature fixtures, so this file restates their PUBLISHED algorithms (funasr
frontends/wav_frontend.py; torchaudio compliance/kaldi.py ``fbank`` with the arguments WavFrontend passes:
num_mel_bins=80, frame_length=25 ms, frame_shift=10 ms, dither, energy_floor=0, window_type="hamming",
sample_frequency=16000, snip_edges=True; defaults preemphasis 0.97, remove_dc_offset, round_to_power_of_two, low_freq 20,
high_freq 0 = Nyquist, use_power, use_log_fbank) and is what the HIP kernel is tested against.  Dither (funasr default 1.0 =
one int16 LSB of Gaussian noise) is taken as 0 here and in the product: deterministic features.
"""
import numpy as np

EPS = np.float32(1.1920928955078125e-07)


def mel_scale(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def mel_banks(num_bins=80, n_fft=512, fs=16000.0, low=20.0, high=0.0):
    """[num_bins, n_fft//2 + 1] triangular filters on the mel scale (last column, the Nyquist bin, is zero)."""
    nyq = 0.5 * fs
    if high <= 0.0:
        high += nyq
    bin_w = fs / n_fft
    mlo, mhi = mel_scale(low), mel_scale(high)
    delta = (mhi - mlo) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = mlo + b * delta, mlo + (b + 1) * delta, mlo + (b + 2) * delta
    mel = mel_scale(bin_w * np.arange(n_fft // 2, dtype=np.float64))[None, :]
    up, down = (mel - left) / (center - left), (right - mel) / (right - center)
    w = np.maximum(0.0, np.minimum(up, down))
    return np.pad(w, ((0, 0), (0, 1))).astype(np.float32)


def fbank(wave, fs=16000, n_mels=80, frame_length_ms=25.0, frame_shift_ms=10.0, preemph=0.97, upscale=True):
    """wave: float32 in [-1, 1) (WavFrontend multiplies by 2^15 first: ``upsacle_samples``) -> [frames, n_mels] float32."""
    x = np.asarray(wave, dtype=np.float32)
    if upscale:
        x = x * np.float32(32768.0)
    win, shift = int(fs * frame_length_ms * 0.001), int(fs * frame_shift_ms * 0.001)
    n_fft = 1 << (win - 1).bit_length()
    if len(x) < win:
        return np.zeros((0, n_mels), dtype=np.float32)
    m = 1 + (len(x) - win) // shift
    idx = np.arange(win)[None, :] + shift * np.arange(m)[:, None]
    fr = x[idx].astype(np.float32)
    fr = fr - fr.mean(1, keepdims=True, dtype=np.float32)                       # remove_dc_offset
    prev = np.concatenate([fr[:, :1], fr[:, :-1]], 1)                           # replicate-padded shift
    fr = fr - np.float32(preemph) * prev
    n = np.arange(win, dtype=np.float64)
    window = (0.54 - 0.46 * np.cos(2.0 * np.pi * n / (win - 1))).astype(np.float32)   # hamming, symmetric
    fr = fr * window[None, :]
    fr = np.pad(fr, ((0, 0), (0, n_fft - win)))
    spec = np.fft.rfft(fr.astype(np.float64), axis=1)
    power = (spec.real ** 2 + spec.imag ** 2).astype(np.float32)
    mel = power @ mel_banks(n_mels, n_fft, float(fs)).T
    return np.log(np.maximum(mel, EPS)).astype(np.float32)


def apply_lfr(feat, lfr_m=7, lfr_n=6):
    """[T, D] -> [ceil(T / n), m * D]: frames i*n .. i*n+m-1 of the sequence left-padded with (m-1)//2 copies of frame 0,
    the tail padded with copies of the last frame."""
    T = feat.shape[0]
    if T == 0:
        return np.zeros((0, lfr_m * feat.shape[1]), dtype=np.float32)
    T_lfr = int(np.ceil(T / lfr_n))
    pad = (lfr_m - 1) // 2
    seq = np.concatenate([np.repeat(feat[:1], pad, 0), feat], 0)
    out = []
    for i in range(T_lfr):
        ids = np.minimum(np.arange(i * lfr_n, i * lfr_n + lfr_m), seq.shape[0] - 1)
        out.append(seq[ids].reshape(-1))
    return np.stack(out).astype(np.float32)


def apply_cmvn(feat, means, scales):
    """(x + means) * scales with the am.mvn <AddShift> / <Rescale> vectors."""
    return ((feat + means[None, : feat.shape[1]]) * scales[None, : feat.shape[1]]).astype(np.float32)


def frontend(wave, means=None, scales=None, lfr_m=7, lfr_n=6, **kw):
    f = apply_lfr(fbank(wave, **kw), lfr_m, lfr_n)
    if means is not None:
        f = apply_cmvn(f, means, scales)
    return f
