"""Shared by oracle/make_golden_dataset.py (which runs the REAL reference dataset on this corpus) and
tests/test_dataset_cpu.py (which runs ps_slm_amd/dataset.py on the same corpus and compares): a tiny deterministic corpus
(wav-in-ark audio, multitask.jsonl, prompt list), a character-level tokenizer and a closed-form stand-in for the audio
front end.  No RNG: everything is a closed-form function of the utterance index so both sides rebuild identical inputs."""
import io
import json
import os
import wave

import numpy as np
import torch

SPEECH_ID, EOS_ID, PAD_ID = 290, 280, 281
N_SAMPLES = [4800, 16000, 7777, 30000, 12345, 5000, 23456, 9000, 16001, 6400, 27000, 8000, 11111]
TARGETS = ["Hello, World 42!", "the quick brown fox", "IT'S 9 o'clock?", "ünïcode stays out", "a", "Mixed CASE, commas, and... dots",
           "  leading space", "numbers 123 456", "what's up", "tab\tinside", "long " * 12, "Z", "last one!"]
GTS = ["hello world", "the quick brown fox", "it's nine o'clock", "caf\\u00e9 society", "", "mixed case", "leading space",
       "one two three", "what's up", "tab inside", "long long", "z", "last one"]
TASKS = ["ASR", "ASR", "hotword", "ASR", "ST", "ASR", "hotword", "ASR", "ST", "ASR", "ASR", "ST", "ASR"]
PROMPTS = [("ASR", "Transcribe the speech."), ("ASR", "Please write down what you hear."), ("ASR", "Recognise: "),
           ("ST", "Translate the speech into German."), ("ST", "Translate: "),
           ("hotword", "Transcribe; pay attention to the words {}."), ("hotword", "Hotwords: {}. Transcribe.")]


class CharTokenizer:
    """encode(): one id per character (code point mod 256), the literal ``<speech>`` -> SPEECH_ID."""
    eos_token_id, pad_token_id, default_ignore_token = EOS_ID, PAD_ID, -100

    def encode(self, text):
        ids, i = [], 0
        while i < len(text):
            if text.startswith("<speech>", i):
                ids.append(SPEECH_ID)
                i += 8
            else:
                ids.append(ord(text[i]) % 256)
                i += 1
        return ids


def waveform_i16(k):
    n = N_SAMPLES[k]
    return (((np.arange(n, dtype=np.int64) * (37 + 2 * k) + 101 * k) % 2001) - 1000).astype(np.int16)


class StandInFrontend:
    """Closed-form stand-in for funasr's WavFrontend (the real one is third-party and absent): T follows the real length
    rule (25 ms / 10 ms frames, LFR n = 6), the 8 'features' per frame are samples picked at fixed strides."""
    fs, win, shift, lfr_n, dim = 16000, 400, 160, 6, 8

    def output_length(self, n):
        frames = 0 if n < self.win else 1 + (n - self.win) // self.shift
        return -(-frames // self.lfr_n)

    def __call__(self, wav):
        wav = np.asarray(wav, dtype=np.float32)
        T = self.output_length(len(wav))
        idx = (np.arange(T)[:, None] * 960 + np.arange(self.dim)[None, :] * 7) % len(wav)
        return torch.from_numpy(wav[idx].astype(np.float32)), T


def write_corpus(root, split_sizes=(("train", 13), ("test", 5))):
    """-> dict(split -> directory).  Audio goes into one Kaldi-style ark of RIFF blobs addressed as ``file.ark:offset``."""
    os.makedirs(root, exist_ok=True)
    ark, offsets = os.path.join(root, "audio.ark"), []
    with open(ark, "wb") as f:
        for k in range(len(N_SAMPLES)):
            key = f"utt{k:02d} ".encode()
            f.write(key)
            offsets.append(f.tell())
            buf = io.BytesIO()
            with wave.open(buf, "wb") as w:
                w.setnchannels(1), w.setsampwidth(2), w.setframerate(16000)
                w.writeframes(waveform_i16(k).tobytes())
            f.write(buf.getvalue())
    with open(os.path.join(root, "multiprompt.jsonl"), "w") as f:
        for task, prompt in PROMPTS:
            f.write(json.dumps({"task": task, "prompt": prompt}) + "\n")
    dirs = {}
    for split, n in split_sizes:
        d = os.path.join(root, split)
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "multitask.jsonl"), "w") as f:
            for k in range(n):
                item = {"key": f"utt{k:02d}", "path": f"{ark}:{offsets[k]}", "target": TARGETS[k], "task": TASKS[k], "GT": GTS[k]}
                if TASKS[k] == "hotword":
                    item["hotword"] = f"fox, clock #{k}"
                f.write(json.dumps(item) + "\n")
        dirs[split] = d
    return dirs


CASES = {   # name -> (split, inference_mode, max_frame_length, world, rank, random seed)
    "train_w1": ("train", False, 400, 1, 0, 0),
    "train_small_budget": ("train", False, 260, 1, 0, 7),
    "train_w2_r0": ("train", False, 400, 2, 0, 3),
    "train_w2_r1": ("train", False, 400, 2, 1, 3),
    "test_infer": ("test", True, 300, 1, 0, 5),
}


def dataset_config(root, dirs, inference_mode, budget):
    from types import SimpleNamespace

    class Cfg(SimpleNamespace):
        def get(self, k, d=None):
            return getattr(self, k, d)

        def __getitem__(self, k):
            return getattr(self, k)

    return Cfg(append_info_tasks=["hotword"], multitask_prompt_path=os.path.join(root, "multiprompt.jsonl"),
               train_scp_file_path=dirs.get("train", ""), dev_scp_file_path=dirs.get("train", ""),
               test_scp_file_path=dirs.get("test", ""), prompt_style="<|im_start|>user\n{}<speech><|im_end|>\n<|im_start|>assistant\n",
               max_audio_length=30, inference_mode=inference_mode, encoder="sensevoice", encoder_path="unused",
               train_max_frame_length=budget, eval_max_frame_length=budget, ds_rate=8)


def flatten_batches(batches):
    """list of collated batches -> flat dict of numpy arrays (npz-friendly); strings as JSON."""
    out = {"n_batches": np.asarray(len(batches))}
    for i, b in enumerate(batches):
        for k, v in b.items():
            if isinstance(v, torch.Tensor):
                out[f"b{i}_{k}"] = v.numpy()
            else:
                out[f"b{i}_{k}"] = np.asarray(json.dumps(v))
    return out
