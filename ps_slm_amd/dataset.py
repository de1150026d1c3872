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
import io
import json
import os
import random
import threading
import re
import struct
import sys
import wave
from functools import partial

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import IterableDataset

# loaded by the reference's SourceFileLoader outside any package (Multitask/utils/dataset_utils.py:14-25): no relative imports
_PKG_PARENT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG_PARENT not in sys.path:
    sys.path.insert(0, _PKG_PARENT)

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


def read_flac(path):
    """``torchaudio.load(x.flac)`` + channel mean (speech_dataset_large.py:123-127) through the native decoder of
    libtasu_hip.so (csrc/flac.hip: host code; every frame CRC and the STREAMINFO MD5 are verified)."""
    import ctypes

    from ps_slm_amd import _lib
    lib = _lib.load()
    blob = np.fromfile(path, dtype=np.uint8)
    info = (ctypes.c_int32 * 3)()
    total = ctypes.c_int64(0)
    rc = lib.tasu_flac_info(blob.ctypes.data, blob.size, info, ctypes.byref(total))
    if rc:
        raise ValueError(f"{path}: not a FLAC stream this decoder reads (tasu_flac_info -> {rc})")
    cap = int(total.value) if total.value else blob.size * 16          # unknown length: generous bound
    out = np.empty(cap, dtype=np.float32)
    n = ctypes.c_int64(0)
    rc = lib.tasu_flac_decode(blob.ctypes.data, blob.size, out.ctypes.data, cap, ctypes.byref(n))
    if rc:
        raise ValueError(f"{path}: FLAC decode failed (tasu_flac_decode -> {rc}: corrupt frame / CRC or MD5 mismatch / unsupported)")
    return int(info[0]), out[: n.value]


def read_audio(path):
    """-> (sample_rate, float32 waveform in [-1, 1)).  ``x.wav`` | ``file.ark:offset`` (kaldiio.load_mat on a wav-in-ark
    entry, speech_dataset_large.py:130-131) | ``x.flac`` (torchaudio.load in the reference, :123-127; here the native decoder behind ``read_flac``)."""
    ext = os.path.splitext(path.split(":")[0])[1].lower()
    if ext == ".flac":
        return read_flac(path)
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
    if os.path.splitext(path)[1].lower() == ".flac":
        import ctypes

        from ps_slm_amd import _lib
        blob = np.fromfile(path, dtype=np.uint8, count=1 << 16)           # STREAMINFO sits at the front
        info, total = (ctypes.c_int32 * 3)(), ctypes.c_int64(0)
        if _lib.load().tasu_flac_info(blob.ctypes.data, blob.size, info, ctypes.byref(total)) == 0 and total.value:
            return int(total.value)
        return len(read_flac(path)[1])
    if ":" in os.path.basename(path):
        fname, off = path.rsplit(":", 1)                                   # wav-in-ark: the RIFF header says it (no sample is read)
        with open(fname, "rb") as f:
            f.seek(int(off))
            with wave.open(f, "rb") as w:
                return w.getnframes()
    with wave.open(path, "rb") as w:
        return w.getnframes()


# ------------------------------------------------------------------------------------------------ dataset
_SPLIT_FIELD = {"train": "train_scp_file_path", "val": "dev_scp_file_path", "test": "test_scp_file_path"}
_TARGET_JUNK = re.compile(r"[^A-Za-z\s.,!?']+")          # what the reference strips from training targets (:164)


def _prompt_table(path):
    """multiprompt.jsonl -> {task: [prompt, ...]} in file order (the order matters: prompts are drawn by index)."""
    table = {}
    with open(path) as f:
        for row in map(json.loads, filter(None, (ln.strip() for ln in f))):
            table.setdefault(row["task"], []).append(row["prompt"])
    return table


def _unescape(text):
    """The jsonl ``GT`` field carries backslash escapes (speech_dataset_large.py:96-102); undecodable text stays as is."""
    try:
        return text.encode("utf-8").decode("unicode_escape")
    except Exception:
        return text


_TOKENIZER_LOCK = threading.Lock()


class MultiTaskDataset(IterableDataset):
    """One sample per ``multitask.jsonl`` line of this rank's (and DataLoader worker's) share."""

    def __init__(self, dataset_config, tokenizer=None, split="train", frontend=None):
        super().__init__()
        if split not in _SPLIT_FIELD:
            raise ValueError("Split must be train val test")
        cfg = dataset_config
        self.dataset_config, self.tokenizer, self.split = cfg, tokenizer, split
        self.data_path = getattr(cfg, _SPLIT_FIELD[split])
        self.multitask_prompt_list = _prompt_table(cfg.multitask_prompt_path)
        self.append_info_tasks = cfg.append_info_tasks
        self.prompt_template = cfg.get("prompt_style", "{}")
        self.inference_mode = bool(cfg.get("inference_mode", False))
        self.text_only = bool(cfg.get("text_only", False))
        self.max_audio_length = cfg.get("max_audio_length", 30)
        self.sample_rate = SAMPLE_RATE
        self.frontend = frontend                    # callable(waveform float32) -> (features [T, D] tensor, T)
        # prompt draws: the GLOBAL ``random`` stream like the reference (speech_dataset_large.py:151) unless the caller hands the
        # split its own generator (``dataset.rng = random.Random(seed)``: the training entrypoint does, so that a reader THREAD and a
        # validation pass on the main thread never interleave draws on one stream -- ADVICE r5)
        self.rng = random
        if frontend is None:
            from ps_slm_amd.frontend import WavFrontend
            self.frontend = WavFrontend.from_encoder_path(cfg.get("encoder_path", None))

    @property
    def _jsonl(self):
        return os.path.join(self.data_path, "multitask.jsonl")

    def __len__(self):
        with open(self._jsonl, "r", encoding="utf-8") as f:
            return sum(1 for _ in f)

    def _share(self):
        """(stride, phase): line n belongs to this iterator when n % stride == phase -- ranks outermost, DataLoader workers
        inside a rank (speech_dataset_large.py:72-91)."""
        info = torch.utils.data.get_worker_info()
        n_work, w = (info.num_workers, info.id) if info is not None else (1, 0)
        distributed = dist.is_available() and dist.is_initialized()
        n_rank, r = (dist.get_world_size(), dist.get_rank()) if distributed else (1, 0)
        return n_work * n_rank, r * n_work + w

    def _audio(self, path, decoded=None):
        if self.text_only:
            return None, self.frontend.output_length(audio_num_samples(path))
        return self.frontend((decoded if decoded is not None else read_audio(path))[1])

    def _sample(self, item, decoded=None):
        task, target = item["task"], item["target"]
        feats, n_frames = self._audio(item["path"], decoded)
        # the prompt is drawn from the task's list with the GLOBAL random stream, after the audio has been read
        text = self.prompt_template.format(self.rng.choice(self.multitask_prompt_list[task]))
        if task in self.append_info_tasks:
            text = text.format(item[task])
        with _TOKENIZER_LOCK:                       # (HF fast tokenizers raise "Already borrowed" under concurrent use: the reader
            head = self.tokenizer.encode(text)      #  thread tokenises the next training batch while a validation pass runs)
        ids = list(head)
        if not self.inference_mode:
            target = _TARGET_JUNK.sub("", target).lower().strip()
            with _TOKENIZER_LOCK:
                ids += self.tokenizer.encode(target) + [self.tokenizer.eos_token_id]
        ids = torch.tensor(ids)
        sample = dict(input_ids=ids, attention_mask=ids.ge(-1), input_features=feats, input_feature_length=n_frames,
                      key=item["key"], target=target, GT=_unescape(item.get("GT", "")))
        if not self.inference_mode:
            lab = ids.clone()
            lab[: len(head)] = self.tokenizer.default_ignore_token
            sample["labels"] = lab
        return sample

    def __iter__(self):
        stride, phase = self._share()
        threads = 0 if self.text_only else int(self.dataset_config.get("decode_threads", 4) or 0)
        with open(self._jsonl) as f:
            mine = (json.loads(line.strip()) for n, line in enumerate(f) if n % stride == phase)
            if threads <= 1:
                for item in mine:
                    yield self._sample(item)
                return
            # The container read + decode of the NEXT utterances (a .flac decodes at ~34 M samples/s per thread in
            # libtasu_hip.so; ctypes, numpy and file reads release the GIL) runs on a few threads ahead of the consumer.  Only
            # that: samples are still built one by one, in file order, on the iterating thread -- the prompt draws from the
            # global ``random`` stream and the front end's launches keep their order.
            from collections import deque
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=threads, thread_name_prefix="tasu-audio") as pool:
                pending = deque()
                for item in mine:
                    pending.append((item, pool.submit(read_audio, item["path"])))
                    if len(pending) > 2 * threads:
                        it, fut = pending.popleft()
                        yield self._sample(it, fut.result())
                while pending:
                    it, fut = pending.popleft()
                    yield self._sample(it, fut.result())

    @staticmethod
    def pad(sequence, max_length, padding_idx=0, padding_style="right"):
        """Tensor padded (or cut) to ``max_length`` along dim 0, filler on the given side."""
        if not isinstance(sequence, torch.Tensor):
            raise TypeError("Type mismatch during padding!")
        missing = max_length - len(sequence)
        if missing <= 0:
            return sequence[:max_length]
        filler = sequence.new_full((missing,) + tuple(sequence.shape[1:]), padding_idx)
        return torch.cat((filler, sequence) if padding_style == "left" else (sequence, filler))

    def collator(self, samples):
        """Batch schema of speech_dataset_large.py:240-305: ids / mask / labels padded to the longest sequence (left in
        inference mode, right in training), features zero-padded along time, GT / keys / targets as lists."""
        assert samples is not None
        side = "left" if self.inference_mode else "right"
        width = max(len(s["input_ids"]) for s in samples)
        tok = self.tokenizer
        stack = lambda key, fill: torch.stack([self.pad(s[key], width, fill, side) for s in samples])
        batch = {"input_ids": stack("input_ids", tok.pad_token_id), "attention_mask": stack("attention_mask", False)}
        batch["input_features"] = None
        if not self.text_only:
            frames = max(s["input_features"].size(0) for s in samples)
            batch["input_features"] = torch.stack(
                [torch.nn.functional.pad(s["input_features"], (0, 0, 0, frames - s["input_features"].size(0))) for s in samples])
        batch["input_feature_length"] = torch.tensor([s["input_feature_length"] for s in samples], dtype=torch.long)
        batch["GT"] = [s["GT"] for s in samples]
        if self.inference_mode:
            batch["keys"], batch["targets"] = [s["key"] for s in samples], [s["target"] for s in samples]
        else:
            batch["labels"] = stack("labels", tok.default_ignore_token)
        return batch


class MultiTaskDynamicBatchDataset(IterableDataset):
    """Pre-batched iterable over ``dataset``: yields lists of samples; ``window_class(sample, pending)`` decides whether
    ``sample`` still fits the pending list (False) or opens a new one (True) -- speech_dataset_large.py:307-330."""

    def __init__(self, dataset, window_class):
        super().__init__()
        if window_class is None:
            raise AssertionError("window_class is required")
        self.dp, self.window_class, self.collator = dataset, window_class, dataset.collator

    def __iter__(self):
        pending = []
        for sample in self.dp:
            if self.window_class(sample, pending):
                if pending:
                    yield pending
                pending = []
            pending.append(sample)
        if pending:
            yield pending

    def __len__(self):
        return len(self.dp)


def window_class(elem, buffer, max_frame_length, ds_rate):
    """True when ``elem`` must open a new batch: an empty buffer (the reference routes the very first element through the
    flush branch as well, :333-335) or batch size x longest merged length over the frame budget (:336-338)."""
    if not buffer:
        return True
    merged = lambda e: len(e["input_ids"]) + e["input_feature_length"] // ds_rate - 1
    longest = max(merged(e) for e in [elem, *buffer])
    return (len(buffer) + 1) * longest > max_frame_length


def get_speech_dataset(dataset_config, tokenizer, split, frontend=None):
    budget = dataset_config.train_max_frame_length if split == "train" else dataset_config.eval_max_frame_length
    fits = partial(window_class, max_frame_length=budget, ds_rate=dataset_config.ds_rate)
    return MultiTaskDynamicBatchDataset(MultiTaskDataset(dataset_config, tokenizer, split, frontend=frontend), fits)
