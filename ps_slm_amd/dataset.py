"""Data side of the path: the reference's jsonl dataset, prompt templating, label construction, collator
and dynamic batching -- ``Multitask/dataset/speech_dataset_large.py`` -- behind the same plugin contract
(``dataset_config.file = "ps_slm_amd/dataset.py:get_speech_dataset"``; loader ``Multitask/utils/dataset_utils.py:28-57``).

What is mirrored, with the reference lines it follows:
  * ``MultiTaskDataset``            jsonl sharding over ranks x workers (:70-91), prompt choice from the task's list with the
                                    global ``random`` module (:151), ``prompt_style`` / ``append_info_tasks`` formatting
                                    (:152-155), target cleaning + EOS + label masking (:162-186), the ``GT`` field (:96-102)
  * ``collator`` / ``pad``          right padding in training, left padding in ``inference_mode`` (:240-305, :188-223)
  * ``MultiTaskDynamicBatchDataset`` + ``window_class``   frame-budget batching (:307-338)
  * ``get_speech_dataset``          split -> max_frame_length (:340-346)

Audio: the reference reads Kaldi ``ark:offset`` entries (wav-in-ark) through kaldiio and FLAC through torchaudio, then runs
funasr's ``WavFrontend`` (80-mel fbank, LFR 7/6, CMVN).  Here ``read_audio`` handles ``.wav`` files and wav-in-ark entries
with the standard library; the features come from ``ps_slm_amd.frontend`` (HIP fbank + LFR + CMVN).  With
``dataset_config.text_only = true`` (the text-only alignment recipe: the model never looks at ``input_features``) only the
audio LENGTH is read -- it decides the batch composition through ``window_class`` -- and no features are computed.
"""
import copy
import io
import json
import os
import random
import re
import struct
import wave
from functools import partial

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import IterableDataset

SAMPLE_RATE = 16000


# ------------------------------------------------------------------------------------------------ audio containers
def _wav_from_fileobj(f):
    with wave.open(f, "rb") as w:
        n, ch, width, rate = w.getnframes(), w.getnchannels(), w.getsampwidth(), w.getframerate()
        raw = w.readframes(n)
    if width != 2:
        raise ValueError(f"only 16-bit PCM wav is supported (sample width {width})")
    x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    if ch > 1:
        x = x.reshape(-1, ch).mean(1)
    return rate, x


def read_audio(path):
    """-> (sample_rate, float32 waveform in [-1, 1)).  ``x.wav`` | ``file.ark:offset`` (kaldiio.load_mat on a wav-in-ark
    entry, speech_dataset_large.py:130-131) | ``x.flac`` (needs torchaudio, absent on this image: a clear error)."""
    ext = os.path.splitext(path.split(":")[0])[1].lower()
    if ext == ".flac":
        raise NotImplementedError("FLAC input needs a decoder (the reference uses torchaudio, speech_dataset_large.py:123-127); "
                                  "convert to wav or wav-in-ark")
    if ":" in os.path.basename(path) or (ext == ".ark" and ":" in path):
        fname, off = path.rsplit(":", 1)
        with open(fname, "rb") as f:
            f.seek(int(off))
            head = f.read(4)
            if head != b"RIFF":
                raise ValueError(f"{path}: not a wav-in-ark entry (the reference takes numpy_array[1] as int16 samples)")
            size = struct.unpack("<I", f.read(4))[0]
            blob = head + struct.pack("<I", size) + f.read(size)
        return _wav_from_fileobj(io.BytesIO(blob))
    with open(path, "rb") as f:
        return _wav_from_fileobj(f)


def audio_num_samples(path):
    """Sample count without decoding (text-only mode)."""
    if ":" in os.path.basename(path):
        return len(read_audio(path)[1])
    with wave.open(path, "rb") as w:
        return w.getnframes()


# ------------------------------------------------------------------------------------------------ dataset
class MultiTaskDataset(IterableDataset):
    def __init__(self, dataset_config, tokenizer=None, split="train", frontend=None):
        super().__init__()
        self.multitask_prompt_list = {}
        self.append_info_tasks = dataset_config.append_info_tasks
        with open(dataset_config.multitask_prompt_path) as f_prompt:
            for line in f_prompt:
                item = json.loads(line.strip())
                self.multitask_prompt_list.setdefault(item["task"], []).append(item["prompt"])
        if split == "train":
            self.data_path = dataset_config.train_scp_file_path
        elif split == "val":
            self.data_path = dataset_config.dev_scp_file_path
        elif split == "test":
            self.data_path = dataset_config.test_scp_file_path
        else:
            raise ValueError("Split must be train val test")
        self.prompt_template = dataset_config.get("prompt_style", "{}")
        self.dataset_config = dataset_config
        self.tokenizer = tokenizer
        self.split = split
        self.max_audio_length = dataset_config.get("max_audio_length", 30)
        self.inference_mode = dataset_config.get("inference_mode", False)
        self.sample_rate = SAMPLE_RATE
        self.text_only = bool(dataset_config.get("text_only", False))
        self.frontend = frontend                    # callable(waveform float32) -> (features [T, D] tensor, T)
        if self.frontend is None:
            from .frontend import WavFrontend
            self.frontend = WavFrontend.from_encoder_path(dataset_config.get("encoder_path", None))

    def __len__(self):
        with open(os.path.join(self.data_path, "multitask.jsonl"), "r", encoding="utf-8") as f:
            return sum(1 for _ in f)

    def __iter__(self):
        worker_info = torch.utils.data.get_worker_info()
        num_workers, worker_id = (1, 0) if worker_info is None else (worker_info.num_workers, worker_info.id)
        if dist.is_available() and dist.is_initialized():
            world_size, rank = dist.get_world_size(), dist.get_rank()
        else:
            world_size, rank = 1, 0
        total_num_workers = num_workers * world_size
        worker_rank = rank * num_workers + worker_id
        with open(os.path.join(self.data_path, "multitask.jsonl")) as f_task:
            for data_index, line in enumerate(f_task):
                if (data_index % total_num_workers) != worker_rank:
                    continue
                item = json.loads(line.strip())
                ark_path, key, target, task = item["path"], item["key"], item["target"], item["task"]
                raw = item.get("GT", "")
                try:
                    GT = raw.encode("utf-8").decode("unicode_escape")
                except Exception:
                    GT = raw
                if self.text_only:
                    input_features = None
                    input_feature_length = self.frontend.output_length(audio_num_samples(ark_path))
                else:
                    _, audio_raw = read_audio(ark_path)
                    input_features, input_feature_length = self.frontend(audio_raw)
                prompt = random.choice(self.multitask_prompt_list[task])
                prompt = self.prompt_template.format(prompt)
                if task in self.append_info_tasks:
                    prompt = prompt.format(item[task])
                prompt_ids = self.tokenizer.encode(prompt)
                prompt_length = len(prompt_ids)
                prompt_ids = torch.tensor(prompt_ids)
                if not self.inference_mode:
                    target = re.sub(r"[^A-Za-z\s.,!?']+", "", target).lower().strip()
                    target_ids = self.tokenizer.encode(target)
                    target_ids.append(self.tokenizer.eos_token_id)
                    input_ids = torch.cat([prompt_ids, torch.tensor(target_ids)])
                else:
                    input_ids = prompt_ids
                result = {"input_ids": input_ids, "attention_mask": input_ids.ge(-1), "input_features": input_features,
                          "input_feature_length": input_feature_length, "key": key, "target": target, "GT": GT}
                if not self.inference_mode:
                    labels = copy.deepcopy(input_ids)
                    labels[:prompt_length] = self.tokenizer.default_ignore_token
                    result["labels"] = labels
                yield result

    @staticmethod
    def pad(sequence, max_length, padding_idx=0, padding_style="right"):
        if not isinstance(sequence, torch.Tensor):
            raise TypeError("Type mismatch during padding!")
        if len(sequence) >= max_length:
            return sequence[:max_length]
        fill = torch.full([max_length - len(sequence)] + list(sequence.size())[1:], padding_idx, dtype=sequence.dtype)
        return torch.cat((sequence, fill)) if padding_style == "right" else torch.cat((fill, sequence))

    def collator(self, samples):
        assert samples is not None
        padding_style = "left" if self.inference_mode else "right"
        L = max(s["input_ids"].shape[0] for s in samples)
        result = {
            "input_ids": torch.stack([self.pad(s["input_ids"], L, self.tokenizer.pad_token_id, padding_style) for s in samples]),
            "attention_mask": torch.stack([self.pad(s["attention_mask"], L, False, padding_style) for s in samples]),
            "input_feature_length": torch.tensor([s["input_feature_length"] for s in samples], dtype=torch.long),
        }
        if self.text_only:
            result["input_features"] = None
        else:
            T = max(s["input_features"].size(0) for s in samples)
            result["input_features"] = torch.stack([
                torch.nn.functional.pad(s["input_features"], (0, 0, 0, T - s["input_features"].size(0)), value=0.0)
                for s in samples])
        result["GT"] = [s["GT"] for s in samples]
        if self.inference_mode:
            result["keys"] = [s["key"] for s in samples]
            result["targets"] = [s["target"] for s in samples]
        else:
            result["labels"] = torch.stack([self.pad(s["labels"], L, self.tokenizer.default_ignore_token, padding_style)
                                            for s in samples])
        return result


class MultiTaskDynamicBatchDataset(IterableDataset):
    """Pre-batched iterable: elements are appended until ``window_class`` says the next one would overflow the budget."""

    def __init__(self, dataset, window_class):
        super().__init__()
        assert window_class is not None
        self.dp, self.window_class, self.collator = dataset, window_class, dataset.collator
        self._buffer = []

    def __iter__(self):
        for elem in self.dp:
            if not self.window_class(elem, self._buffer):
                self._buffer.append(elem)
            else:
                if len(self._buffer) > 0:
                    yield self._buffer
                self._buffer = [elem]
        if len(self._buffer) > 0:
            yield self._buffer
        self._buffer = []

    def __len__(self):
        return len(self.dp)


def window_class(elem, buffer, max_frame_length, ds_rate):
    """speech_dataset_large.py:333-338 -- NOTE the empty-buffer case returns True, so the very first element is routed
    through the 'flush' branch of the batcher (which has nothing to flush) exactly as in the reference."""
    if len(buffer) == 0:
        return True
    frames = lambda e: len(e["input_ids"]) + (e["input_feature_length"] // ds_rate) - 1
    max_frame = max(frames(elem), max(frames(b) for b in buffer))
    return (len(buffer) + 1) * max_frame > max_frame_length


def get_speech_dataset(dataset_config, tokenizer, split, frontend=None):
    dataset = MultiTaskDataset(dataset_config, tokenizer, split, frontend=frontend)
    budget = dataset_config.train_max_frame_length if split == "train" else dataset_config.eval_max_frame_length
    return MultiTaskDynamicBatchDataset(dataset, partial(window_class, max_frame_length=budget, ds_rate=dataset_config.ds_rate))
