"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/dataset_*.npz by running the REAL reference dataset
(/root/reference/Multitask/dataset/speech_dataset_large.py: MultiTaskDataset, collator, MultiTaskDynamicBatchDataset,
window_class, get_speech_dataset) on the tiny corpus of tests/dataset_fixtures.py.

The reference module imports whisper / kaldiio / torchaudio / funasr and builds its front end through
SenseVoiceSmall.from_pretrained; none of those third-party packages is installed here, so stand-in modules (OUR code,
names only) are injected: kaldiio.load_mat reads the wav-in-ark entry, funasr's extract_fbank calls the closed-form
StandInFrontend of tests/dataset_fixtures.py.  What the fixtures therefore pin is everything the reference itself owns:
jsonl sharding over ranks, prompt choice through the global ``random`` stream, prompt templating / hotword insertion,
target cleaning, EOS, label masking, GT unescaping, right/left padding, feature padding and frame-budget batching.
Feature extraction is NOT pinned by these fixtures (see oracle/fbank_oracle.py).

Run in the build container only:  python oracle/make_golden_dataset.py
"""
import importlib.util
import io
import os
import random
import struct
import sys
import tempfile
import types
import wave

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dataset_fixtures as fx  # noqa: E402

REFERENCE_ROOT = os.environ.get("TASU_REFERENCE_ROOT", "/root/reference")
FRONTEND = fx.StandInFrontend()


def _load_mat(path):
    fname, off = path.rsplit(":", 1)
    with open(fname, "rb") as f:
        f.seek(int(off))
        head = f.read(8)
        size = struct.unpack("<I", head[4:])[0]
        blob = head + f.read(size)
    with wave.open(io.BytesIO(blob), "rb") as w:
        return w.getframerate(), np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")


def _extract_fbank(audio_list, data_type="sound", frontend=None):
    feats, lens = zip(*(frontend(np.asarray(a)) for a in audio_list))
    return list(feats), list(lens)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference_dataset():
    _mod("whisper")
    _mod("kaldiio", load_mat=_load_mat)
    ta = _mod("torchaudio")
    _mod("torchaudio.compliance")
    ta.compliance = sys.modules["torchaudio.compliance"]
    ta.compliance.kaldi = _mod("torchaudio.compliance.kaldi")
    _mod("funasr")
    _mod("funasr.utils")
    _mod("funasr.utils.load_utils", load_audio_text_image_video=lambda data, **kw: data, extract_fbank=_extract_fbank)

    class _SenseVoiceSmall:
        @staticmethod
        def from_pretrained(path):
            return object(), {"frontend": FRONTEND}

    _mod("model")
    _mod("model.SenseVoice", SenseVoiceSmall=_SenseVoiceSmall)
    path = os.path.join(REFERENCE_ROOT, "Multitask", "dataset", "speech_dataset_large.py")
    spec = importlib.util.spec_from_file_location("ref_speech_dataset_large", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    import torch.distributed as dist
    ref = load_reference_dataset()
    out_dir = os.path.join(ROOT, "tests", "golden")
    with tempfile.TemporaryDirectory() as root:
        dirs = fx.write_corpus(root)
        for name, (split, infer, budget, world, rank, seed) in fx.CASES.items():
            cfg = fx.dataset_config(root, dirs, infer, budget)
            saved = dist.is_initialized, dist.get_world_size, dist.get_rank
            if world > 1:
                dist.is_initialized, dist.get_world_size, dist.get_rank = (lambda: True), (lambda *a: world), (lambda *a: rank)
            try:
                random.seed(seed)
                ds = ref.get_speech_dataset(cfg, fx.CharTokenizer(), split)
                batches = [ds.collator(raw) for raw in ds]
                n = len(ds)
            finally:
                dist.is_initialized, dist.get_world_size, dist.get_rank = saved
            flat = fx.flatten_batches(batches)
            flat["dataset_len"] = np.asarray(n)
            path = os.path.join(out_dir, f"dataset_{name}.npz")
            np.savez_compressed(path, **flat)
            print(f"dataset_{name}: {len(batches)} batches, sizes {[int(b['input_ids'].shape[0]) for b in batches]}, "
                  f"{os.path.getsize(path) / 1024:.1f} KB")


if __name__ == "__main__":
    main()
