"""funasr ``WavFrontend`` on the MI355X: waveform -> Kaldi log-mel fbank -> low-frame-rate stacking -> CMVN, the features
``SenseVoiceSmall`` eats (``[T, 560]`` at 60 ms per frame).  The reference runs this on the host inside its dataset
(Multitask/dataset/speech_dataset_large.py:133-146: ``load_audio_text_image_video`` + ``extract_fbank(frontend=...)`` with
the frontend that ``SenseVoiceSmall.from_pretrained`` builds from ``<encoder_path>/config.yaml`` ``frontend_conf`` and
``am.mvn``).  Here the two kernels of ``csrc/frontend.hip`` do it on the device; constants (window, mel matrix, CMVN
vectors) are built once on the host with numpy.  Dither is 0 (deterministic features; funasr's default adds one int16 LSB of
noise).  PARITY UNPINNED against funasr/torchaudio (absent): oracle/fbank_oracle.py restates the published algorithm.
"""
import math
import os

import numpy as np
import torch


def _mel(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def mel_matrix(n_mels, n_fft, fs, low=20.0, high=0.0):
    """torchaudio.compliance.kaldi.get_mel_banks without VTLN: [n_mels, n_fft//2 + 1] (zero Nyquist column)."""
    high = high + 0.5 * fs if high <= 0.0 else high
    mlo, mhi = _mel(low), _mel(high)
    delta = (mhi - mlo) / (n_mels + 1)
    b = np.arange(n_mels, dtype=np.float64)[:, None]
    left, center, right = mlo + b * delta, mlo + (b + 1) * delta, mlo + (b + 2) * delta
    mel = _mel(fs / n_fft * np.arange(n_fft // 2, dtype=np.float64))[None, :]
    w = np.maximum(0.0, np.minimum((mel - left) / (center - left), (right - mel) / (right - center)))
    return np.pad(w, ((0, 0), (0, 1))).astype(np.float32)


def load_cmvn(path):
    """Kaldi-nnet ``am.mvn``: the vector after <AddShift>/<LearnRateCoef> is the shift (negated means), the one after
    <Rescale>/<LearnRateCoef> the scale (funasr frontends/wav_frontend.py load_cmvn)."""
    means = scales = None
    with open(path, "r", encoding="utf-8") as f:
        lines = f.readlines()
    for i, line in enumerate(lines):
        parts = line.split()
        if not parts:
            continue
        if parts[0] in ("<AddShift>", "<Rescale>"):
            nxt = lines[i + 1].split()
            if nxt[0] == "<LearnRateCoef>":
                vec = np.array([float(v) for v in nxt[3:len(nxt) - 1]], dtype=np.float32)
                if parts[0] == "<AddShift>":
                    means = vec
                else:
                    scales = vec
    if means is None or scales is None:
        raise ValueError(f"{path}: no <AddShift>/<Rescale> vectors found")
    return means, scales


class WavFrontend:
    def __init__(self, fs=16000, n_mels=80, frame_length=25, frame_shift=10, lfr_m=7, lfr_n=6, cmvn=None, window="hamming",
                 ops=None, device="cuda"):
        if window != "hamming":
            raise NotImplementedError("WavFrontend window types other than 'hamming' (SenseVoiceSmall's) are not implemented")
        self.fs, self.n_mels, self.lfr_m, self.lfr_n = fs, n_mels, lfr_m, lfr_n
        self.win, self.shift = int(fs * frame_length * 0.001), int(fs * frame_shift * 0.001)
        if self.win > 512:
            raise NotImplementedError("frames longer than 512 samples need a longer FFT than the kernel's")
        self.device = torch.device(device)
        self._ops = ops
        n = np.arange(self.win, dtype=np.float64)
        self._window = torch.from_numpy((0.54 - 0.46 * np.cos(2.0 * np.pi * n / (self.win - 1))).astype(np.float32))
        self._mel = torch.from_numpy(mel_matrix(n_mels, 512, float(fs)))
        self._means = self._scales = None
        if cmvn is not None:
            self._means, self._scales = (torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for v in cmvn)
        self._on_device = False

    @classmethod
    def from_encoder_path(cls, encoder_path, **kw):
        """``<encoder_path>/config.yaml`` ``frontend_conf`` + ``am.mvn`` when they exist, SenseVoiceSmall's values otherwise."""
        conf, cmvn = {}, None
        cfg = os.path.join(str(encoder_path or ""), "config.yaml")
        if os.path.isfile(cfg):
            import yaml
            conf = (yaml.safe_load(open(cfg)) or {}).get("frontend_conf", {}) or {}
        mvn = conf.get("cmvn_file") or "am.mvn"
        for cand in (mvn, os.path.join(str(encoder_path or ""), os.path.basename(mvn))):     # as written, else next to config.yaml
            if os.path.isfile(cand):
                cmvn = load_cmvn(cand)
                break
        args = {k: conf[k] for k in ("fs", "n_mels", "frame_length", "frame_shift", "lfr_m", "lfr_n", "window") if k in conf}
        return cls(cmvn=cmvn, **args, **kw)

    # frames of the fbank and of the stacked output for an n-sample waveform (what decides the batch composition)
    def num_frames(self, n_samples):
        return 0 if n_samples < self.win else 1 + (n_samples - self.win) // self.shift

    def output_length(self, n_samples):
        return int(math.ceil(self.num_frames(n_samples) / self.lfr_n))

    @property
    def ops(self):
        if self._ops is None:
            from .ops import HipOps
            self._ops = HipOps()
        return self._ops

    def __call__(self, waveform):
        """waveform: float32 numpy / tensor in [-1, 1) at ``fs`` -> (features [T, lfr_m * n_mels] on the device, T)."""
        w = torch.as_tensor(np.ascontiguousarray(waveform, dtype=np.float32) if not isinstance(waveform, torch.Tensor) else waveform,
                            dtype=torch.float32).to(self.device).contiguous()
        if not self._on_device:
            self._window, self._mel = self._window.to(self.device), self._mel.to(self.device)
            if self._means is not None:
                self._means, self._scales = self._means.to(self.device), self._scales.to(self.device)
            self._on_device = True
        n = int(w.numel())
        T = self.num_frames(n)
        if T == 0:
            return torch.zeros(0, self.lfr_m * self.n_mels, device=self.device), 0
        fb = torch.empty(T, self.n_mels, dtype=torch.float32, device=self.device)
        self.ops.fbank(w, n, 32768.0, self.win, self.shift, self._window, self._mel, self.n_mels, 0.97, fb)
        T_lfr = self.output_length(n)
        out = torch.empty(T_lfr, self.lfr_m * self.n_mels, dtype=torch.float32, device=self.device)
        self.ops.lfr_cmvn(fb, T, self.n_mels, self.lfr_m, self.lfr_n, self._means, self._scales, out)
        return out, T_lfr
