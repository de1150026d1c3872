"""Throughput of the REAL data path: the training ENTRYPOINT
(ps_slm_amd.finetune_deepspeed.main) at full Qwen2.5-1.5B / SenseVoiceSmall geometry on a generated corpus of 30-second
utterances -- jsonl -> ps_slm_amd/dataset.py (wav-in-ark through the standard library / .flac through the native decoder) ->
HIP fbank + LFR + CMVN -> collate -> dynamic batching -> the step -- next to bench.py's synthetic-input figures.

  config 2 (text-only CPS, dataset_config.text_only=true: only the audio LENGTH is read)   vs bench.py's headline
  config 4 (audio-SFT: the waveforms are read, decoded, turned into features on the device) vs bench.py's audio_sft (S = 628)

The corpus lives in tmpfs (/dev/shm when writable): N entries of one ark file, 16 utterances per batch by the frame budget.
Reference being replaced: Multitask/finetune_deepspeed.py:185-208 (4 DataLoader worker processes),
Multitask/dataset/speech_dataset_large.py:120-146.  Prints one JSON object."""
import argparse
import io
import json
import os
import shutil
import sys
import tempfile
import time
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SECONDS, RATE = 30, 16000


def make_corpus(root, n_utts, kind="ark", n_distinct=32, seed=7):
    """multitask.jsonl + prompt list + audio.  Every utterance: 24 prompt ids + <speech>, 127 target ids (+ EOS), a 104-word GT
    (the pseudo-posterior's ids), 30 s of audio -> the benchmark's S = 256 in text-only mode."""
    rng = np.random.default_rng(seed)
    os.makedirs(root, exist_ok=True)
    n = SECONDS * RATE
    t = np.arange(n) / RATE
    waves = []
    for k in range(n_distinct):                                     # band-limited noise + a few tones: speech-like dynamics
        x = rng.standard_normal(n).astype(np.float32)
        x = np.convolve(x, np.ones(8, dtype=np.float32) / 8, mode="same") * (0.3 + 0.7 * np.abs(np.sin(2 * np.pi * (0.5 + k % 5) * t)))
        x += 0.2 * np.sin(2 * np.pi * (180 + 30 * k) * t).astype(np.float32)
        waves.append(np.clip(x * 6000, -32000, 32000).astype("<i2"))
    paths = []
    if kind == "ark":
        ark = os.path.join(root, "audio.ark")
        with open(ark, "wb") as f:
            for i in range(n_utts):
                f.write(f"utt{i:05d} ".encode())
                off = f.tell()
                buf = io.BytesIO()
                with wave.open(buf, "wb") as w:
                    w.setnchannels(1), w.setsampwidth(2), w.setframerate(RATE)
                    w.writeframes(waves[i % n_distinct].tobytes())
                f.write(buf.getvalue())
                paths.append(f"{ark}:{off}")
    else:
        import flac_fixtures as ff                                  # (a slow pure-Python writer: a handful of distinct files)
        files = []
        for k in range(min(n_distinct, 4)):
            p = os.path.join(root, f"utt{k}.flac")
            with open(p, "wb") as f:
                f.write(ff.write_flac(waves[k].astype(np.int64), rate=RATE, bps=16, blocksize=4096, seed=k))
            files.append(p)
        paths = [files[i % len(files)] for i in range(n_utts)]
    with open(os.path.join(root, "multiprompt.jsonl"), "w") as f:
        f.write(json.dumps({"task": "ASR", "prompt": " ".join(str(100 + j) for j in range(24))}) + "\n")
    d = os.path.join(root, "train")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "multitask.jsonl"), "w") as f:
        for i in range(n_utts):
            tgt = " ".join(chr(97 + int(v)) for v in rng.integers(0, 26, 127))          # letters survive the target cleaning
            gt = " ".join(str(int(v)) for v in rng.integers(1, 25055, 104))
            f.write(json.dumps({"key": f"utt{i:05d}", "path": paths[i], "target": tgt, "task": "ASR", "GT": gt}) + "\n")
    return d


def run(root, train_dir, text_only, epochs, workers, graphs=True, budget=4200):
    from ps_slm_amd.finetune_deepspeed import main
    argv = ["++model_config.file=ps_slm_amd/ps_slm.py:model_factory", "++model_config.llm_path=synthetic:qwen2.5-1.5b",
            "++model_config.llm_dim=1536", "++model_config.encoder_dim=25055", "++model_config.encoder_projector=linear-silu",
            "++train_config.freeze_llm=true", "++train_config.freeze_encoder=true", "++train_config.use_fp16=true",
            f"++train_config.gt_emb={'true' if text_only else 'false'}", "++train_config.gt_emb_noise=false",
            "++train_config.ctc_posterior=true", "++train_config.do_psd=true", f"++train_config.num_epochs={epochs}",
            f"++train_config.num_workers_dataloader={workers}", "++train_config.run_validation=false",
            "++dataset_config.file=ps_slm_amd/dataset.py:get_speech_dataset", f"++dataset_config.train_scp_file_path={train_dir}",
            f"++dataset_config.multitask_prompt_path={root}/multiprompt.jsonl", "++dataset_config.prompt_style={} 151665",
            f"++dataset_config.train_max_frame_length={budget}", "++dataset_config.ds_rate=5",
            f"++dataset_config.text_only={'true' if text_only else 'false'}", "++metric=acc", "++log_config.log_interval=1000000",
            f"++use_graphs={'true' if graphs else 'false'}"]
    t0 = time.perf_counter()
    res = main(argv)
    return {"epoch_utterances_per_s": [round(v, 1) for v in res["epoch_utterances_per_s"]], "steps": res["steps"],
            "wall_s_incl_model_build": round(time.perf_counter() - t0, 1)}


def main_cli():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=2048)
    ap.add_argument("--flac-utts", type=int, default=256)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--no-flac", action="store_true")
    ap.add_argument("--no-inline", action="store_true", help="skip the num_workers_dataloader=0 legs (everything on the training thread)")
    args = ap.parse_args()
    base = "/dev/shm" if os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    root = tempfile.mkdtemp(prefix="tasu_corpus_", dir=base)
    out = {"corpus": {"utterances": args.utts, "seconds_each": SECONDS, "where": base, "format": "wav-in-ark (one file, RIFF entries)"}}
    try:
        t0 = time.perf_counter()
        tr = make_corpus(os.path.join(root, "ark"), args.utts, "ark")
        out["corpus"]["build_s"] = round(time.perf_counter() - t0, 1)
        out["text_only_reader_thread"] = run(os.path.join(root, "ark"), tr, True, args.epochs, 1)
        out["audio_wav_reader_thread"] = run(os.path.join(root, "ark"), tr, False, args.epochs, 1)
        if not args.no_inline:
            out["audio_wav_inline"] = run(os.path.join(root, "ark"), tr, False, args.epochs, 0)
        if not args.no_flac:
            t0 = time.perf_counter()
            fr = make_corpus(os.path.join(root, "flac"), args.flac_utts, "flac")
            out["corpus"]["flac_build_s"] = round(time.perf_counter() - t0, 1)
            out["audio_flac_reader_thread"] = run(os.path.join(root, "flac"), fr, False, args.epochs, 1)
    finally:
        shutil.rmtree(root, ignore_errors=True)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main_cli()
