#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: train utterances/sec of the TASU alignment step
(SenseVoiceSmall -> projector -> Qwen2.5-1.5B, text-only CPS recipe, bf16) at N GPUs of one node.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over one synthetic batch per GPU: host merge plan + tiny H2D of the
integer inputs, pseudo-posterior build, projector fwd, 28-layer decoder fwd, lm_head + CE, dgrad-only backward,
projector wgrad, gradient all-reduce over RCCL (N > 1) and fused AdamW.  Weights are seeded random-init at the
exact Qwen2.5-1.5B / projector geometry (no pretrained weights exist on the box).  The frozen SenseVoice encoder
pass, whose result the reference throws away in text-only mode (Multitask/model/ps-slm.py:430-454 vs :459-468),
is NOT executed (stated in config.workload).

Prints ONE JSON line (rank 0) with `roofline` (MFMA GEMM kernel, timed live with HIP events on the launch
stream) and, at N = 1, `cpu_baseline` (the oracle's CPU port timed on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ps_slm_amd.streams import ensure_hw_queues, report as stream_report     # noqa: E402

ensure_hw_queues()                     # before the first HIP call: a hardware queue each for the main stream and the side-stream roles

PMC_FILE = "r06_gemm_pmc.json"          # this round's counter passes (tools/make_round_artifacts.sh PART=pmc -> tools/pmc_shapes.py)
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA ~2.5 PF dense"


def gemm_flops_per_utt(geo, S, n_audio, n_head_rows=None, tail_rows=False):
    """FLOPs that run inside the MFMA GEMM kernel per utterance (weights only, multiply-add =
    2 FLOPs): decoder linears + lm_head, forward and dgrad-only backward, projector fwd + bwd.  ``n_head_rows``: positions
    the lm_head is EXECUTED on (the training step projects only the positions that carry a label); None = all S, which
    is SURVEY's algorithmic figure.  ``tail_rows``: the step also ran the LAST decoder layer's MLP (gate|up, down and their
    dgrads) on those rows only (TasuModel.tail_rows): the other S - n_head_rows rows of that layer's MLP are not executed and
    not counted."""
    D, I, H, G, V, L = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads, geo.llm_vocab, geo.llm_layers
    per_layer = D * (H + 2 * G) * 128 + H * 128 * D + 3 * D * I
    proj_tok = 2 * (geo.ctc_vocab * geo.bottleneck + geo.bottleneck * D)
    head_rows = S if n_head_rows is None else n_head_rows
    skipped = 2 * 2 * 3 * D * I * (S - head_rows) if tail_rows else 0
    return 2 * (2 * L * per_layer * S + 2 * V * D * head_rows) + 3 * proj_tok * n_audio - skipped


def total_flops_per_utt(geo, S, n_audio, n_head_rows=None, tail_rows=False):
    """SURVEY.md 8d text-only total (adds causal attention: fwd 2*S*D*L per token, bwd 2.5x)."""
    attn = geo.llm_layers * 2 * S * geo.llm_heads * 128 * S
    return gemm_flops_per_utt(geo, S, n_audio, n_head_rows, tail_rows) + attn + 2.5 * attn


class TimedOps:
    """Wraps the GEMM launches of HipOps (gemm and the gate|up GEMM with the SwiGLU epilogue: every launch whose FLOPs
    gemm_flops_per_utt counts) with HIP event pairs recorded on the launch stream (torch's current stream)."""

    def __init__(self, ops):
        self._ops = ops
        self.enabled = False
        self.events = []
        self.recording = False         # collect the GEMM calls of a step (for the GEMM-only graph replay)
        self.calls = []
        self.replay_ms = None          # (ms per replay of the recorded calls, number of calls)
        self.kernel_launches = None    # GEMM kernel launches of the recorded step (tasu_gemm_launch_count difference)

    def __getattr__(self, name):
        return getattr(self._ops, name)

    def _timed(self, fn, a, k):
        if self.recording:
            self.calls.append((fn, a, k))
        if not self.enabled:
            return fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(*a, **k)
        e1.record()
        self.events.append((e0, e1))

    def gemm(self, *a, **k):
        return self._timed(self._ops.gemm, a, k)

    def gemm_gate_up_swiglu(self, *a, **k):
        return self._timed(self._ops.gemm_gate_up_swiglu, a, k)

    def gemm_splitk(self, *a, **k):
        return self._timed(self._ops.gemm_splitk, a, k)

    def gemm_qkv_rope(self, *a, **k):
        return self._timed(self._ops.gemm_qkv_rope, a, k)

    def gemm_dswiglu(self, dy, wd_t, gu, dgu, dact_ws, M, I, K):
        # tasu_gemm_dswiglu = the GEMM (timed) + tasu_swiglu_bwd (not a GEMM: not timed) -- the same two launches
        self.gemm(dy, wd_t, dact_ws, M, I, K)
        return self._ops.swiglu_bwd(dact_ws, gu, dgu, M, I)

    def time_replay(self, reps):
        """The recorded GEMM calls of one step, captured in launch order as ONE hipGraph and replayed: the sum of the GEMM launch
        durations without host gaps (an eager event pair around a C call that launches two kernels also times the host between
        them).  Same kernels, shapes and buffers as the step; operands are whatever the last step left in them."""
        import gc
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        gc.disable()                                       # (no cyclic collection inside a capture: TasuModel._graphed)
        try:
            with torch.cuda.graph(g):
                for fn, a, k in self.calls:
                    fn(*a, **k)
        finally:
            gc.enable()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        self.replay_ms = (e0.elapsed_time(e1) / reps, len(self.calls))
        del g

    def total_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.events)


def pooled_state_dict(geo, seed=1234):
    """Full-geometry reference-named CPU weights cut from one seeded 64M-element N(0, 0.02) pool (drawing 1.5 G
    fresh normals costs ~40 s of host time and the baseline only needs representative, non-degenerate values)."""
    from ps_slm_amd.synthetic import random_state_dict

    g = torch.Generator().manual_seed(seed)
    pool = torch.randn(1 << 26, generator=g) * 0.02
    small = random_state_dict(type(geo).from_dict(dict(vars(geo), llm_layers=0, llm_vocab=8)), seed, with_encoder=False)
    sd, off = {}, 0

    def cut(*shape):
        nonlocal off
        n = 1
        for d in shape:
            n *= d
        if off + n > pool.numel():
            off = (off * 7 + 13) % 4099
        if n > pool.numel():
            t = pool.repeat((n + pool.numel() - 1) // pool.numel())[:n].clone()
        else:
            t = pool[off:off + n].clone()
        off += n
        return t.view(*shape)

    D, I, H, G, V = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads, geo.llm_vocab
    sd["llm.model.embed_tokens.weight"] = cut(V, D)
    for l in range(geo.llm_layers):
        p = f"llm.model.layers.{l}."
        sd[p + "input_layernorm.weight"] = torch.ones(D)
        sd[p + "post_attention_layernorm.weight"] = torch.ones(D)
        for nm, shp in (("q_proj", (H * 128, D)), ("k_proj", (G * 128, D)), ("v_proj", (G * 128, D))):
            sd[p + f"self_attn.{nm}.weight"] = cut(*shp)
            sd[p + f"self_attn.{nm}.bias"] = cut(shp[0])
        sd[p + "self_attn.o_proj.weight"] = cut(D, H * 128)
        sd[p + "mlp.gate_proj.weight"] = cut(I, D)
        sd[p + "mlp.up_proj.weight"] = cut(I, D)
        sd[p + "mlp.down_proj.weight"] = cut(D, I)
    sd["llm.model.norm.weight"] = torch.ones(D)
    for k, v in small.items():
        if k.startswith("encoder_projector."):
            sd[k] = v
    return sd


_SD_CACHE = {}


def cpu_baseline_worker(kinds):
    for kind in kinds.split(","):
        cpu_baseline_one(kind)


def cpu_baseline_one(kind):
    """Runs in a CHILD process (no GPU): the oracle (CPU port of the reference path, oracle/tasu_oracle.py) at FULL
    geometry, fp32, on a BOUNDED sample of the GPU workload (training at B = 1 and B = 16, decode at
    B = 1 and B = 16).  kind: train1 | train16 | decode1 | decode16.  Encoder pass skipped like the GPU leg."""
    import dataclasses

    from oracle import tasu_oracle as O
    from ps_slm_amd.model import Geometry
    from ps_slm_amd.synthetic import synthetic_text_batch

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(cores, 64))
    torch.set_num_threads(threads)
    geo = Geometry.qwen25_1p5b()
    sd = _SD_CACHE.setdefault("sd", None) or _SD_CACHE.update(sd=pooled_state_dict(geo)) or _SD_CACHE["sd"]     # (built once per worker)
    gd = dataclasses.asdict(geo)
    host = f"torch {torch.__version__}, {threads} threads ({cores} usable cores), encoder pass skipped"
    if kind.startswith("train"):
        B = int(kind[5:])
        batch = synthetic_text_batch(geo, B, seed=1234)
        m = {k: torch.zeros_like(sd[k]) for k in O.PROJ_KEYS}
        v = {k: torch.zeros_like(sd[k]) for k in O.PROJ_KEYS}
        times, t_start = [], time.perf_counter()
        budget = 25.0 if B == 1 else 20.0                        # (B = 16: ONE iteration of ~40 s on 64 threads is the bounded sample)
        for step in range(1, 5):
            t0 = time.perf_counter()
            out, grads = O.loss_and_projector_grads(sd, batch, gd, "fp32")
            for k in O.PROJ_KEYS:
                O.adamw_step(sd[k], grads[k], m[k], v[k], step, 5e-5)
            times.append(time.perf_counter() - t0)
            if step >= (2 if B == 1 else 1) and time.perf_counter() - t_start > budget:
                break
        timed = times[1:] if len(times) > 1 else times           # B = 16: a single iteration may already fill the budget
        print(json.dumps({"value": round(B / (sum(timed) / len(timed)), 4), "unit": "utterances/s", "cores": threads, "kind": "port",
                          "sample": f"oracle/tasu_oracle.py fp32, B={B} utterance(s) (S=256, 104 audio tokens), fwd+bwd+AdamW, "
                                    f"{len(timed)} timed iteration(s) after {len(times) - len(timed)} warm-up"
                                    + (" (ONE COLD iteration: ~40 s each, the bounded sample has no room for a warm-up)"
                                       if len(times) == len(timed) else "") + f", {host}"}), flush=True)
        return
    B = int(kind[6:])
    new = 24 if B == 1 else 6                                    # bounded sample: prompt pass + this many cached positions
    batch = synthetic_text_batch(geo, B, seed=1234, noise=False)
    ids = batch["input_ids"][:, :25]
    am = torch.ones_like(ids, dtype=torch.bool)
    post, plen = O.pseudo_posterior(batch["post_ids"], geo.ctc_vocab)
    with torch.no_grad():
        emb, mask, _, _ = O.merge(O.projector(sd, post, "fp32"), plen, sd["llm.model.embed_tokens.weight"][ids], ids, am, None,
                                  geo.speech_id)
        stamps = [time.perf_counter()]
        toks = O.beam_search_generate(sd, emb, mask, gd, num_beams=4, max_new_tokens=new, eos_token_id=-1, pad_token_id=0,
                                      kv_cache=True, step_times=stamps)
    # HF generate's algorithm (prompt pass once, then one cached position per step: oracle qwen2_hidden_step).  The GPU leg emits 200
    # positions per utterance; the CPU sample is the prompt pass + `new` positions, priced at the GPU leg's length:
    # 200 B / (t_prompt + 199 t_position)
    t_prompt = stamps[1] - stamps[0]
    t_pos = (stamps[-1] - stamps[1]) / max(len(stamps) - 2, 1)
    print(json.dumps({"value": round(B * 200 / (t_prompt + 199 * t_pos), 3), "unit": "tokens/s", "cores": threads, "kind": "port",
                      "prompt_pass_s": round(t_prompt, 2), "s_per_position": round(t_pos, 4),
                      "sample": f"oracle beam search fp32 WITH KV cache (HF generate's algorithm), B={B}, beam 4, prompt 128 + "
                                f"{toks.shape[1]} positions timed, priced at 200 positions: 200 B / (t_prompt + 199 t_position), {host}"}),
          flush=True)


def cpu_baselines(kinds, timeout_s=300):
    """{kind: record} of the oracle timed in ONE child process (no GPU) that builds the fp32 weights once and runs the kinds one after
    the other on all cores it is given (side by side they distort each other 3-5x: measured)."""
    import subprocess
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    unit = lambda k: "utterances/s" if k.startswith("train") else "tokens/s"
    out = {}
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", ",".join(kinds)], capture_output=True,
                           text=True, timeout=timeout_s, env=env, cwd=ROOT)
        recs = [json.loads(ln) for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        for k, rec in zip(kinds, recs):
            out[k] = rec
        for k in kinds[len(recs):]:
            out[k] = {"value": None, "unit": unit(k), "cores": 0, "kind": "port", "sample": f"{k} worker failed: " + r.stderr[-300:]}
    except subprocess.TimeoutExpired as e:
        so = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        recs = [json.loads(ln) for ln in so.strip().splitlines() if ln.startswith("{")]
        for k, rec in zip(kinds, recs):
            out[k] = rec
        for k in kinds[len(recs):]:
            out[k] = {"value": None, "unit": unit(k), "cores": 0, "kind": "port", "sample": f"{k} worker exceeded {timeout_s}s on this host"}
    return out


def cpu_baseline(kind="train1", timeout_s=240):
    return cpu_baselines([kind], timeout_s)[kind]


def decode_leg(core, raw, B, new_tokens=200, beams=4):
    """Second half of BASELINE.json's metric ("...; decode tok/s"): beam-4 generate of the same model on rank 0, prompt of
    24 ids + <speech> -> 128 merged positions, exactly `new_tokens` generated positions per utterance (an EOS id that never
    matches keeps every run the same length).  Timed region = prefill + the whole decode loop, second run."""
    from ps_slm_amd.decode import beam_search_generate
    ids = raw["input_ids"][:, :25]
    am = torch.ones_like(ids, dtype=torch.bool)

    def run():
        st = core.prepare_text(ids, am, None, raw["post_ids"], None, None)
        core.forward_projector_text(st)
        return beam_search_generate(core, st, num_beams=beams, max_new_tokens=new_tokens, eos_token_id=-1, pad_token_id=0)

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_new = int(out.shape[1])
    geo = core.geo
    weight_bytes = 2 * sum(int(w[k].numel()) for w in core.llm.layers for k in ("wqkv", "wo", "wgu", "wd")) \
        + 2 * int(core.llm.head.numel())
    # K/V bytes a position reads: every beam row attends over its whole context (28,672 B x context x rows at 1.5B);
    # averaged over the generated positions (context 128 .. 128 + n_new - 1)
    kv_row_pos = geo.llm_layers * 2 * geo.llm_kv_heads * 128 * 2
    kv_bytes = kv_row_pos * (128 + (n_new - 1) / 2.0) * B * beams
    per_pos = dt / n_new
    return {"metric": "decode tokens/sec (beam 4, emitted tokens)", "value": round(B * n_new / dt, 1), "unit": "tokens/s",
            "beam_tokens_per_s": round(beams * B * n_new / dt, 1), "ms_per_step": round(per_pos * 1e3, 3),
            "config": {"utterances": B, "beams": beams, "prefill_len": 128, "new_tokens": n_new},
            "roofline": {"bound": "hbm", "achieved": round((weight_bytes + kv_bytes) / per_pos / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round((weight_bytes + kv_bytes) / per_pos / 8e12, 4),
                         "weight_bytes_per_position": weight_bytes, "kv_bytes_per_position_avg": int(kv_bytes),
                         "note": "(bf16 weight bytes streamed once + K/V bytes read by the 64 beam rows) per generated position / "
                                 "wall time per position (prefill included in the wall time)"}}


def decode_fp32_leg(local_rank, B, new_tokens=200, beams=4, model_name="qwen2.5-1.5b"):
    """The reference's OWN decode arithmetic (train_config.use_fp16 = false: fp32 weights, cache and logits; Multitask/inference_batch.py:
    113-117) on the fp32 path (ps_slm_amd/decode_fp32.py): same prompt, beams and length as decode_leg."""
    from ps_slm_amd.config import ModelConfig, TrainConfig
    from ps_slm_amd.decode_fp32 import beam_search_generate_fp32
    from ps_slm_amd.ps_slm import model_factory
    from ps_slm_amd.synthetic import synthetic_text_batch

    tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, gt_emb_noise=False, ctc_posterior=True, do_psd=True, use_fp16=False,
                     batching_strategy="dynamic")
    mc = ModelConfig(llm_path=f"synthetic:{model_name}", encoder_projector="linear-silu", encoder_dim=25055,
                     llm_dim=3584 if model_name == "qwen2.5-7b" else 1536)
    model, _ = model_factory(tc, mc, device=f"cuda:{local_rank}", init_seed=1234, keep_logits=False, with_encoder=False)
    core = model.core
    geo = core.geo
    raw = synthetic_text_batch(geo, B, seed=1234, noise=False)
    ids = raw["input_ids"][:, :25]
    am = torch.ones_like(ids, dtype=torch.bool)

    def run():
        st = core.prepare_text(ids, am, None, raw["post_ids"], None, None)
        return beam_search_generate_fp32(core, st, num_beams=beams, max_new_tokens=new_tokens, eos_token_id=-1, pad_token_id=0)

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_new = int(out.shape[1])
    weight_bytes = 4 * (sum(int(f[k].numel()) for f in core.llm.f32["layers"] for k in ("wqkv", "wo", "wgu", "wd")) + int(core.llm.f32["head"].numel()))
    kv_bytes = geo.llm_layers * 2 * geo.llm_kv_heads * 128 * 4 * (128 + (n_new - 1) / 2.0) * B * beams
    per_pos = dt / n_new
    rec = {"metric": "decode tokens/sec (beam 4, emitted tokens, fp32 arithmetic: use_fp16=false)", "value": round(B * n_new / dt, 1),
           "unit": "tokens/s", "ms_per_step": round(per_pos * 1e3, 3),
           "config": {"utterances": B, "beams": beams, "prefill_len": 128, "new_tokens": n_new, "dtype": "f32"},
           "roofline": {"bound": "hbm", "achieved": round((weight_bytes + kv_bytes) / per_pos / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round((weight_bytes + kv_bytes) / per_pos / 8e12, 4), "weight_bytes_per_position": weight_bytes,
                        "kv_bytes_per_position_avg": int(kv_bytes),
                        "note": "(fp32 weight bytes streamed once + fp32 K/V bytes read by the 64 beam rows) per generated position / wall "
                                "time per position (fp32 prefill included in the wall time)"}}
    core._dec_graphs.clear()
    del model, core
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return rec


def train_fp32_leg(local_rank, B, steps=3):
    """The reference's SHIPPED training arithmetic (train_config.use_fp16 = false: forward and backward outside autocast) on the fp32
    path (ps_slm_amd/train_fp32.py): the headline batch, a few steps, through TasuEngine.  A correctness mode: reported so that
    nobody has to guess what it costs."""
    from ps_slm_amd.config import DEFAULT_DS_CONFIG, ModelConfig, TrainConfig, load_ds_config
    from ps_slm_amd.engine import TasuEngine
    from ps_slm_amd.ps_slm import model_factory
    from ps_slm_amd.synthetic import synthetic_text_batch

    tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, gt_emb_noise=False, ctc_posterior=True, do_psd=True, use_fp16=False,
                     batching_strategy="dynamic")
    mc = ModelConfig(llm_path="synthetic:qwen2.5-1.5b", encoder_projector="linear-silu", encoder_dim=25055, llm_dim=1536)
    model, _ = model_factory(tc, mc, device=f"cuda:{local_rank}", init_seed=1234, keep_logits=False, with_encoder=False)
    eng = TasuEngine(model, load_ds_config(DEFAULT_DS_CONFIG))
    eng.train()
    raw = synthetic_text_batch(model.core.geo, B, seed=1234, noise=False)
    batch = dict(input_ids=raw["input_ids"], attention_mask=raw["attention_mask"], labels=raw["labels"], input_features=None,
                 input_feature_length=None, GT=[" ".join(map(str, p)) for p in raw["post_ids"]])

    def step():
        out, _ = eng(**batch)
        eng.backward(out.loss)
        eng.step()
        return out

    step()                                                  # warm-up: allocations, the transposed fp32 weight copies
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    geo = model.core.geo
    st = eng._last_state
    flops = total_flops_per_utt(geo, st.S, st.Ra / B) * B
    rec = {"metric": "train utterances/sec (fp32 arithmetic: use_fp16=false, the reference's shipped recipe)", "value": round(B / dt, 2),
           "unit": "utterances/s", "ms_per_step": round(dt * 1e3, 2),
           "config": {"per_gpu_batch": B, "seq_len": st.S, "dtype": "f32", "final_loss": round(float(out.loss.detach()), 4)},
           "roofline": {"bound": "mfma", "achieved": round(flops / dt / 1e12, 1), "peak": 157.3, "unit": "TFLOP/s",
                        "frac": round(flops / dt / 1e12 / 157.3, 4),
                        "note": "SURVEY 8d's algorithmic FLOPs of the step (all positions through the lm_head: the fp32 step has no "
                                "labelled-rows shortcut) against the fp32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md)"}}
    eng.destroy()
    del eng, model
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return rec


def encoder_gemm_flops_per_utt(geo, frames):
    """GEMM FLOPs of the frozen SANM encoder + CTC head per utterance (forward only; SURVEY 8d: 2 x 3,145,728 per frame-layer
    for the 512-wide layers, layer 0 reads 560 features, CTC head 2 x 512 x 25055 per frame; attention not counted here)."""
    E, Ff, Fd, V = geo.enc_dim, geo.enc_ffn, geo.feat_dim, geo.ctc_vocab
    layers = geo.enc_blocks + geo.enc_tp_blocks
    per_layer = 3 * E * E + E * E + 2 * E * Ff
    first = 3 * E * Fd + E * E + 2 * E * Ff
    return 2 * frames * ((layers - 1) * per_layer + first + E * V)


def lora_flops_per_utt(geo, S, lp):
    """The adapters' FLOPs of one utterance, split by where they run: (inside the timed base GEMMs, elsewhere).  Inside: the rank
    columns of the K-extended forward GEMMs, 2 S N (kext - K) per adapted group (padding included: executed work).  Elsewhere
    (tasu_gemm_nt_rank, tasu_lora_apply): forward us = x A^T (2 S r in), backward du, dx, dB, dA (2 S r (in + out) each pair)."""
    L = geo.llm_layers
    ext = L * sum(2 * S * lp.nout[g] * (lp.kext[g] - lp.kbase[g]) for g, _ in lp.groups)
    rank = L * sum(2 * S * lp.r * lp.dims[t][0] + 2 * 2 * S * lp.r * (lp.dims[t][0] + lp.dims[t][1]) for t in lp.cfg.target_modules)
    return ext, rank


def train_leg(args, model_name, path, B, steps, warmup, world, rank, local_rank, want_decode, device=None, ops=None,
              variable=False, blank_biased=False, lora=False, force_exchange=False):
    """One training workload (path: "text" = the text-only CPS recipe of configs 2/3/5, "audio" = config 4: 500 feature frames
    through the SenseVoice encoder, CTC posterior, PSD, projector, LLM).  Returns the fields of a bench record.
    ``device`` / ``ops``: tests/test_bench_cpu.py drives this function over gloo with the CPU operator double (the N > 1
    bookkeeping -- barriers, MAX-reduce of the wall time, rank-0-only record, exposed all-reduce time -- without a GPU).
    ``variable``: SURVEY 8d's variable-S run -- 8 distinct batches, CPS token drop 0.05 (fresh draws every step), shapes padded to
    the training entrypoint's buckets (``++graph_buckets=16,8,256``) so that the LRU of captured step graphs gets hits: the
    throughput of real dynamic batching, where no two steps need have the same shape.  ``blank_biased`` (audio path): the CTC
    head's blank bias is raised until PSD keeps ~100 of the 500 frames (a trained encoder's regime; the random-init posterior
    keeps ~476).  ``force_exchange`` (N = 1): the chunked gradient exchange runs all the same, through a ONE-rank RCCL communicator
    -- every all-reduce launch, event chain and wait of the N > 1 step with nothing on the wire: a sanity figure for
    ``allreduce_exposed_ms`` from hardware."""
    import torch.distributed as dist

    from ps_slm_amd.config import DEFAULT_DS_CONFIG, ModelConfig, TrainConfig, load_ds_config
    from ps_slm_amd.engine import TasuEngine
    from ps_slm_amd.ps_slm import model_factory
    from ps_slm_amd.synthetic import synthetic_text_batch

    audio = path == "audio"
    train_config = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=not audio, gt_emb_noise=not audio, ctc_posterior=True,
                               do_psd=True, use_fp16=True, batching_strategy="dynamic")
    train_config.use_peft = bool(lora)                  # the reference's defaults: r = 64, alpha = 16, dropout 0.05, all 7 Linears
    model_config = ModelConfig(llm_path=f"synthetic:{model_name}", encoder_projector="linear-silu", encoder_dim=25055,
                               llm_dim={"qwen2.5-1.5b": 1536, "qwen2.5-7b": 3584, "mid": 256}[model_name])
    device = device or f"cuda:{local_rank}"
    on_gpu = device.startswith("cuda")
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    model, _ = model_factory(train_config, model_config, device=device, init_seed=1234, keep_logits=False,
                             with_encoder=audio, **({"ops": ops} if ops is not None else {}))
    model.drop_prob = 0.05 if variable else args.drop_prob
    core = model.core
    core.use_graphs = on_gpu and not args.no_graphs
    if variable:
        core.shape_buckets = (16, 8, 256)
    timed = TimedOps(core.ops)
    core.ops = timed
    engine = TasuEngine(model, load_ds_config(DEFAULT_DS_CONFIG), force_exchange=force_exchange)
    engine.train()
    exchanging = world > 1 or (force_exchange and engine.exchange)
    geo = core.geo
    def make_batch(seed, **shape):
        raw = synthetic_text_batch(geo, B, seed=seed, noise=False, **shape)
        if audio:
            return raw, dict(input_ids=raw["input_ids"], attention_mask=raw["attention_mask"], labels=raw["labels"],
                             input_features=raw["input_features"], input_feature_length=raw["input_feature_length"], GT=None)
        GT = [" ".join(map(str, p)) for p in raw["post_ids"]]
        return raw, dict(input_ids=raw["input_ids"], attention_mask=raw["attention_mask"], labels=raw["labels"],
                         input_features=None, input_feature_length=None, GT=GT)

    raw, batch = make_batch(1234 + rank)
    # variable shapes: 8 ragged batches of different audio / target lengths (dynamic batching gives every batch its own shape)
    var_shapes = [(104, 128), (96, 120), (88, 101), (104, 90), (73, 128), (99, 111), (80, 84), (92, 125)]
    batches = [batch] if not variable else [make_batch(1234 + rank + 100 * i, n_audio=na, target_len=tl, ragged=True)[1]
                                             for i, (na, tl) in enumerate(var_shapes)]
    torch.manual_seed(1234 + rank)          # CPS alpha / keep draws come from the global CPU RNG, like the reference
    blank_note = None
    if blank_biased:
        # bisection on the blank logit's bias: forward passes only, until PSD keeps 90-110 rows per utterance
        enc = core.encoder
        lo_b, hi_b, bias0 = 0.0, 30.0, float(enc.ctc_b[geo.blank_id])
        if getattr(args, "blank_bias", None) is not None:
            lo_b = hi_b = float(args.blank_bias)
        for _ in range(12):
            mid_b = 0.5 * (lo_b + hi_b)
            enc.ctc_b[geo.blank_id] = bias0 + mid_b
            try:
                engine(**batch)
                kept = engine._last_state.Ra / B
            except ValueError:                      # "PSD removed every frame": the bias is far too high
                kept = 0.0
            if 90 <= kept <= 110:
                break
            lo_b, hi_b = (mid_b, hi_b) if kept > 110 else (lo_b, mid_b)
        blank_note = f"CTC blank bias raised by {mid_b:.2f}: PSD keeps {kept:.0f} rows per utterance (padded to the batch maximum)"
    seen_shapes = set()
    step_no = [0]
    ahead = audio and not getattr(args, "no_encoder_ahead", False)
    ahead_hits = [0]

    def step():
        b = batches[step_no[0] % len(batches)]
        step_no[0] += 1
        out, acc = engine(**b)
        if ahead:                                       # the next batch's frozen encoder pass under this batch's decoder step
            ahead_hits[0] += bool(engine.prefetch(**batches[step_no[0] % len(batches)]))
        seen_shapes.add((engine._last_state.S, engine._last_state.Ra))
        engine.backward(out.loss)
        engine.step()
        return out

    for _ in range(warmup):
        step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    # hipGraph replay cannot carry per-launch event pairs, so with graphs the GEMM launches are timed in a second,
    # eager pass over the same number of steps right after the timed region (same kernels, same shapes, same data).
    timed.enabled = on_gpu and rank == 0 and not core.use_graphs
    lib0 = getattr(core.ops, "lib", None)
    k_eager0 = int(lib0.tasu_gemm_launch_count()) if (timed.enabled and lib0 is not None) else None
    engine.time_exchange = exchanging
    engine.exposed_events = []
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if k_eager0 is not None:                            # eager launches: the GEMM kernel launches of the timed steps themselves
        timed.kernel_launches = (int(lib0.tasu_gemm_launch_count()) - k_eager0) // max(steps, 1)
    timed.enabled = False
    exposed_ms = engine.exposed_ms() / max(steps, 1) if exchanging else 0.0
    engine.time_exchange = False
    if world > 1:
        t = torch.tensor([dt, exposed_ms], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, exposed_ms = float(t[0].item()), float(t[1].item())
    if core.use_graphs:
        # hipGraph replay cannot carry per-launch event pairs: one eager step records the step's GEMM calls (every rank takes
        # part: the step contains the gradient all-reduce), rank 0 then replays exactly those launches as a GEMM-only graph
        core.use_graphs = False
        step()
        sync()
        timed.recording = rank == 0
        lib = getattr(core.ops, "lib", None)
        k0 = int(lib.tasu_gemm_launch_count()) if lib is not None else 0
        step()
        sync()
        timed.recording = False
        # kernel launches behind this step's GEMM calls (a column-split call is two; lm_head split-K adds its own) -- counted by the
        # library itself; the encoder's GEMMs of an audio step that runs one batch ahead are launched by the same step() call
        timed.kernel_launches = (int(lib.tasu_gemm_launch_count()) - k0) if lib is not None else None
        if rank == 0:
            timed.time_replay(steps)
        core.use_graphs = True
    loss = float(out.loss.detach())
    rec = None
    if rank == 0:
        st = engine._last_state
        S = st.S
        n_audio = st.Ra / B                             # projector rows per utterance (audio: PSD output, padded to the batch max)
        n_head = st.nLp / B                             # lm_head rows executed per utterance (labelled positions, padded to 64)
        enc = encoder_gemm_flops_per_utt(geo, raw["input_features"].shape[1] + 4) if audio else 0
        tail = "xout_tail" in st.dev                    # last layer's MLP ran on the labelled rows only (TasuModel.tail_rows)
        lo_in, lo_out = lora_flops_per_utt(geo, S, core.lora) if core.lora is not None else (0, 0)
        gemm_flops_step = (gemm_flops_per_utt(geo, S, n_audio, n_head, tail) + enc + lo_in) * B      # the timed GEMM calls only
        executed_step = (total_flops_per_utt(geo, S, n_audio, n_head, tail) + enc + lo_in + lo_out) * B
        survey_step = (total_flops_per_utt(geo, S, n_audio) + enc + lo_in + lo_out) * B
        if timed.replay_ms is not None:
            gemm_ms, n_launch = timed.replay_ms[0] * steps, timed.replay_ms[1] * steps
        else:
            gemm_ms, n_launch = timed.total_ms(), len(timed.events)
        achieved = gemm_flops_step * steps / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        k_launch = timed.kernel_launches if timed.kernel_launches else n_launch // max(steps, 1)
        # HBM-side bytes per GEMM call come from the round's own PMC passes (counters need their own rocprofv3 --pmc runs, FETCH_SIZE and
        # WRITE_SIZE apart: tools/make_round_artifacts.sh PART=pmc): the call-count-weighted mean over the step's GEMM shapes, cold
        # rotating operands, gfx950 x2 fetch correction.  Only for the configuration those passes ran.
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", PMC_FILE)
        if os.path.isfile(pmc) and model_name == "qwen2.5-1.5b" and B == 16 and not audio and not variable and core.lora is None:
            from ps_slm_amd._lib import gemm_source_hash
            rec_pmc = json.load(open(pmc))
            if rec_pmc.get("gemm_source_hash") == gemm_source_hash():     # the counters describe the kernels that just ran
                traffic = rec_pmc.get("traffic_bytes_per_launch")
                traffic_src = (f"bytes per GEMM call, profiles/{PMC_FILE} (gemm_source_hash {rec_pmc['gemm_source_hash']} = this library's "
                               "kernels): separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/gemm_shapes_run.py (the step's "
                               "GEMM shapes, cold rotating operands), fetch x2 (gfx950 tallies 16-B/lane streaming reads at half), "
                               "call-count-weighted over one step; fabric side of the L2s (Infinity-Cache hits included)")
            else:
                traffic_src = (f"null: profiles/{PMC_FILE} was collected on other GEMM kernel sources (its gemm_source_hash "
                               f"{rec_pmc.get('gemm_source_hash')} != {gemm_source_hash()}); re-run tools/make_round_artifacts.sh PART=pmc")
        what = ("audio-SFT step (500 feature frames -> SANM encoder -> CTC posterior -> PSD -> projector -> LLM fwd+dgrad bwd+"
                "projector wgrad+AdamW)" if audio else
                "text-only CPS alignment step (fwd+dgrad bwd+projector wgrad+AdamW), frozen encoder pass skipped")
        if variable:
            what += (f"; VARIABLE shapes: 8 ragged batches (73-104 audio tokens, 84-128 target tokens per utterance), CPS token drop "
                     f"0.05 redrawn every step, shapes padded to buckets (16 token columns, 8 posterior rows, 256 labelled rows): "
                     f"{len(seen_shapes)} distinct (S, posterior rows) seen, {len(core._graphs)} step graphs captured; value = "
                     f"utterances / wall time over all shapes; S, FLOPs and roofline describe the LAST step's shape only")
        if blank_note:
            what += "; " + blank_note
        if audio and ahead_hits[0]:
            what += ("; the frozen encoder runs ONE BATCH AHEAD on a side stream (every step = the encoder pass of batch i + 1 under "
                     "the PSD / projector / LLM step of batch i; same kernels, same results: TasuModel.prefetch_encoder)")
        if core.lora is not None:
            c = core.lora.cfg
            what += (f"; LoRA recipe (use_peft=true): r={c.r}, alpha={c.lora_alpha:g}, dropout {c.lora_dropout:g} on {len(c.target_modules)} "
                     f"Linears per layer, {core.lora.num_parameters() / 1e6:.1f} M adapter parameters trained next to the projector "
                     f"(decoder wgrads + AdamW over the larger bucket in the step)")
        rec = {
            "value": round(world * B * steps / dt, 2), "unit": "utterances/s", "ms_per_step": round(dt / steps * 1e3, 3),
            "config": {"workload": f"{what}, {model_name}, {B} utterances/GPU x S={S} "
                                   f"({n_audio:.0f} audio tokens/utterance), lm_head + CE on the {st.nL} labelled positions of the "
                                   f"batch only (the other rows' loss and gradient are identically zero)",
                       "per_gpu_batch": B, "seq_len": S, "parallelism": f"dp{world}", "final_loss": round(loss, 4)},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 1), "peak": MFMA_BF16_DENSE_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel": "tasu_pp::gemm_pp_kernel + tasu_pipe::gemm_pipe_kernel (tasu_gemm_nt_bf16_ws, tasu_gemm_gate_up_swiglu, tasu_gemm_qkv_rope, tasu_gemm_dswiglu, tasu_gemm_nt_bf16_splitk)",
                         # launches = KERNEL launches (what rocprofv3's per-kernel average counts); calls = C-ABI GEMM calls
                         "launches_per_step": k_launch, "calls_per_step": n_launch // max(steps, 1),
                         "avg_launch_us": round(gemm_ms * 1e3 / max(steps, 1) / max(k_launch, 1), 2),
                         "algorithmic_gflop_per_launch": round(gemm_flops_step / max(k_launch, 1) / 1e9, 2),
                         "gemm_ms_per_step": round(gemm_ms / max(steps, 1), 3),
                         "launch": "hipGraph replay" if core.use_graphs else "eager",
                         "passes_note": ("value / ms_per_step: hipGraph replay of the timed steps; achieved / avg_launch_us: the step's "
                                         "GEMM calls (recorded from one eager step: same kernels, shapes, buffers, order) replayed as a "
                                         "GEMM-only hipGraph between two HIP events, per call (a call of the column-split policy is two "
                                         "kernel launches); graphs cannot carry per-launch events, and eager event pairs also time the "
                                         "host between the two launches of a split call.  In this replay the GEMMs follow each other "
                                         "without the kernels that produce their operands in between; rocprofv3's sum over the same "
                                         "launches of an eager step agrees within 2 % (profiles/r03_bench_summary.md)") if core.use_graphs else
                                        "one eager pass, HIP event pairs around every GEMM call (includes the host gap inside two-launch calls)",
                         "whole_step_tflops": round(executed_step * steps / dt / 1e12, 1),     # per GPU (executed_step counts one rank's batch)
                         "whole_step_frac": round(executed_step * steps / dt / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
                         "whole_step_frac_at_survey_flops": round(survey_step * steps / dt / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
                         "flops_note": "achieved / whole_step_frac count EXECUTED FLOPs per GPU (lm_head rows without a label, and the "
                                       "last decoder layer's MLP on those rows, are not computed and not counted); "
                                       "whole_step_frac_at_survey_flops prices the same utterances/s at SURVEY 8d's algorithmic "
                                       "FLOPs, which count them"},
        }
        if exchanging:
            rec["collective"] = engine.comm_info()      # ranks = ncclCommCount of the communicator the exchange ran on
            rec["allreduce_exposed_ms"] = round(exposed_ms, 3)
            rec["allreduce_note"] = (f"per step, max over ranks: time the compute stream waited for gradient ranges in step() "
                                     f"(event pairs around every wait); bucket exchanged in {len(core.grad_ranges(engine.w1_chunks))} ranges.  Expected "
                                     f"on 8 xGMI-connected GPUs: ~0.35 ms (the last 51-MB row block of the Linear1 weight gradient at "
                                     f"the ~300 GB/s bus bandwidth RCCL reaches; the whole 218-MB bucket would be ~1.3 ms), DESIGN.md 6")
        if on_gpu and (core.lora is not None or audio or exchanging):
            rec["side_streams"] = stream_report(device)   # which hardware queue every overlap role got (ps_slm_amd/streams.py)
        if want_decode:
            rec["decode"] = decode_leg(core, raw, B, new_tokens=200 if model_name != "qwen2.5-7b" else 64)
    engine.destroy()                                    # RCCL communicator, before the process group goes
    core._graphs.clear()                                # this leg's captured graphs go NOW, with nothing capturing
    getattr(core, "_dec_graphs", {}).clear()
    del engine, model, core, timed
    import gc
    gc.collect()
    if on_gpu:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return rec


SUB_METRICS = {"audio_sft": " (audio-SFT, config 4)", "qwen2.5-7b": " (Qwen2.5-7B align, config 5)",
               "variable_S": " (text-only, variable shapes: CPS drop 0.05, bucketed hipGraphs)",
               "audio_sft_blank_biased": " (audio-SFT, config 4, ~100 audio tokens per utterance)",
               "lora_r64": " (text-only, use_peft=true: LoRA r=64 on the decoder + projector)",
               "exchange_1rank": " (headline step with the N > 1 gradient exchange forced through a 1-rank RCCL "
                                 "communicator: allreduce_exposed_ms is the sanity figure)",
               "decode_fp32": None, "train_fp32": None}                     # (carries its own metric: decode tokens/s in the reference's fp32 arithmetic)

STDOUT_LINE_LIMIT = 8000           # the driver's record keeps ~8,000 characters of stdout and parses the line from them
_ROOF_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches_per_step", "calls_per_step",
              "avg_launch_us", "gemm_ms_per_step", "launch", "whole_step_frac", "whole_step_frac_at_survey_flops")


def _clip(s, n=120):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 3] + "..."


def _brief(r):
    """value / ms_per_step / roofline fractions of a sub-record."""
    if not isinstance(r, dict) or r.get("value") is None:
        return None
    out = {"value": r["value"], "unit": r.get("unit"), "ms_per_step": r.get("ms_per_step")}
    roof = r.get("roofline")
    if isinstance(roof, dict):
        out["roofline_frac"] = roof.get("frac")
        if "whole_step_frac_at_survey_flops" in roof:
            out["whole_step_frac_at_survey_flops"] = roof["whole_step_frac_at_survey_flops"]
    if "allreduce_exposed_ms" in r:
        out["allreduce_exposed_ms"] = r["allreduce_exposed_ms"]
    return out


def compact_line(full, full_path="profiles/bench_r06_full.json"):
    """The ONE stdout line: the contract's top-level keys, ``roofline`` and ``cpu_baseline`` as plain numbers and short names, and
    ``digest`` (value / ms_per_step / roofline fractions of every sub-record) -- no prose, every string <= 120 characters, the whole
    line < STDOUT_LINE_LIMIT bytes.  The complete record (sub-record bodies, notes, sources) goes to stderr and to ``full_path``."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: full.get(k) for k in top}
    cfg = full.get("config") or {}
    model = "Qwen2.5-7B" if "7B" in str(full.get("metric")) else "Qwen2.5-1.5B"
    what = "audio-SFT step" if "audio-SFT" in str(cfg.get("workload")) else "text-only CPS alignment step"
    if "LoRA recipe" in str(cfg.get("workload")):
        what += " + LoRA r=64"
    line["config"] = {"workload": _clip(f"{what} (fwd + dgrad bwd + projector wgrad + AdamW), {model}, "
                                        f"{cfg.get('per_gpu_batch')} utt/GPU x S={cfg.get('seq_len')}"),
                      "per_gpu_batch": cfg.get("per_gpu_batch"), "seq_len": cfg.get("seq_len"), "parallelism": cfg.get("parallelism"),
                      "final_loss": cfg.get("final_loss")}
    roof = full.get("roofline") or {}
    line["roofline"] = {k: _clip(roof[k]) for k in _ROOF_KEYS if k in roof}
    if "kernel" in line["roofline"]:
        line["roofline"]["kernel"] = "gemm_pp_kernel + gemm_pipe_kernel (all MFMA GEMM launches of the step)"
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": _clip(_short_sample(cb.get("sample")))}
    if "allreduce_exposed_ms" in full:
        coll = full.get("collective") or {}
        line["collective"] = {"ranks": coll.get("ranks"), "library": _clip(coll.get("library"), 60)} if isinstance(coll, dict) else None
        line["allreduce_exposed_ms"] = full["allreduce_exposed_ms"]
    digest = {"train_1p5b" if "7B" not in str(full.get("metric")) else "train_7b": _brief(full), "decode": _brief(full.get("decode"))}
    for name in SUB_METRICS:
        rec = full.get(name)
        if isinstance(rec, dict):
            digest[name] = _brief(rec) or {"error": _clip(rec.get("error"))}
            if isinstance(rec.get("decode"), dict):
                digest[name + "_decode"] = _brief(rec["decode"])
    dp = full.get("data_path")
    if isinstance(dp, dict):
        digest["data_path"] = ({k: dp.get(k) for k in ("text_only_utterances_per_s", "audio_wav_utterances_per_s")} if "error" not in dp
                               else {"error": _clip(dp["error"])})
    for k, rec in (full.get("cpu_baselines") or {}).items():
        if isinstance(rec, dict):
            digest["cpu_" + k] = {"value": rec.get("value"), "unit": rec.get("unit"), "cores": rec.get("cores")}
    line["digest"] = {k: v for k, v in digest.items() if v is not None}
    if "wall_seconds" in full:
        line["wall_seconds"] = full["wall_seconds"]
    line["full_record"] = full_path
    return line


def _short_sample(s):
    """cpu_baseline.sample in <= 120 characters: what was timed, batch, iterations, threads."""
    if not isinstance(s, str):
        return s
    import re
    m = re.search(r"B=(\d+).*?(\d+) timed iteration\(s\) after (\d+) warm-up.*?(\d+) threads", s)
    if m:
        return f"oracle port fp32, B={m.group(1)}, S=256, fwd+bwd+AdamW, {m.group(2)} timed iter after {m.group(3)} warm-up, {m.group(4)} threads"
    return s


def emit(full, fd, full_path):
    """Full record -> stderr + ``full_path`` (best effort); compact line (< STDOUT_LINE_LIMIT bytes, checked) -> the real stdout."""
    text = json.dumps(full)
    sys.stderr.write("FULL_RECORD " + text + "\n")
    sys.stderr.flush()
    try:
        os.makedirs(os.path.dirname(os.path.join(ROOT, full_path)), exist_ok=True)
        with open(os.path.join(ROOT, full_path), "w") as f:
            f.write(text + "\n")
    except OSError:
        pass
    out = json.dumps(compact_line(full, full_path))
    if len(out.encode()) >= STDOUT_LINE_LIMIT:            # (cannot happen with the fixed key set; never print an unparsable line)
        slim = compact_line(full, full_path)
        slim.pop("wall_seconds", None)
        slim["digest"] = {k: v for k, v in slim["digest"].items() if k.startswith(("train", "decode"))}
        out = json.dumps(slim)
    os.write(fd, (out + "\n").encode())


def check_launch(world, gpus):
    """``--gpus N`` must be launched as N processes (torch.distributed.run, one per GPU): anything else would print a line whose
    n_gpus does not describe the run."""
    if world != gpus:
        if world == 1:
            raise SystemExit(f"--gpus {gpus}: launch multi-GPU runs with torch.distributed.run (one process per GPU): python -m "
                             f"torch.distributed.run --nnodes=1 --nproc-per-node {gpus} --master-addr 127.0.0.1 bench.py --gpus {gpus}")
        raise SystemExit(f"--gpus {gpus} but WORLD_SIZE is {world}: the launcher's process count and --gpus must agree")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="utterances per GPU per step")
    ap.add_argument("--model", default="qwen2.5-1.5b", choices=["qwen2.5-1.5b", "qwen2.5-7b", "mid"])
    ap.add_argument("--path", default="text", choices=["text", "audio"], help="headline workload: text-only CPS or audio-SFT")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--drop-prob", type=float, default=0.0, help="CPS token drop (0 keeps S fixed at 256)")
    ap.add_argument("--no-graphs", action="store_true", help="launch every kernel eagerly instead of hipGraph replay")
    ap.add_argument("--no-decode", action="store_true", help="skip the decode tok/s leg (second half of BASELINE.json's metric)")
    ap.add_argument("--blank-biased", action="store_true",
                    help="--path audio: raise the CTC blank bias until PSD keeps ~100 frames per utterance (a trained encoder's regime)")
    ap.add_argument("--blank-bias", type=float, default=None,
                    help="--path audio --blank-biased: use this bias instead of searching for it (profiled runs: no search passes in the trace)")
    ap.add_argument("--no-encoder-ahead", action="store_true",
                    help="audio path: run every batch's frozen encoder pass in front of its own step instead of one batch ahead on a side stream")
    ap.add_argument("--lora", action="store_true", help="the use_peft=true recipe (LoRA r=64 on the decoder's 7 Linears) as the measured workload")
    ap.add_argument("--no-extra", action="store_true", help="skip the config-4 (audio-SFT) and config-5 (Qwen2.5-7B) sub-records")
    ap.add_argument("--no-data-path", action="store_true",
                    help="skip the data_path sub-record (the training entrypoint on a generated 2048-utterance wav-in-ark corpus in tmpfs)")
    ap.add_argument("--full-record", default="profiles/bench_r06_full.json",
                    help="where the complete record (every sub-record's body and notes) is written, relative to the repo; stdout carries "
                         "only the compact line")
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args.cpu_baseline_worker)

    import torch.distributed as dist

    # ONE JSON line on stdout, whatever the libraries print: RCCL writes a version banner to the C stdout at communicator
    # creation, which C stdio flushes at process exit -- AFTER the JSON line.  File descriptor 1 is therefore pointed at stderr for
    # the whole run and the line goes to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    check_launch(world, args.gpus)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))

    wall, t_mark = {}, [time.perf_counter()]

    def lap(name):                                   # wall seconds of each leg of this run (model builds and graph captures included)
        now = time.perf_counter()
        wall[name] = round(now - t_mark[0], 1)
        t_mark[0] = now
    main_rec = train_leg(args, args.model, args.path, args.batch, args.steps, args.warmup, world, rank, local_rank,
                         want_decode=world == 1 and not args.no_decode and args.path == "text",
                         blank_biased=args.blank_biased and args.path == "audio", lora=args.lora)
    lap("headline" + ("+decode" if "decode" in main_rec else ""))
    extras = {}
    headline = args.model == "qwen2.5-1.5b" and args.path == "text" and not args.lora
    if world == 1 and headline and not args.no_extra:
        # BASELINE.json configs 4 and 5 as sub-records of the same line (shorter runs: their steps are 3-4x longer)
        extras["variable_S"] = train_leg(args, "qwen2.5-1.5b", "text", args.batch, 32, 24, 1, 0, local_rank, False, variable=True)
        lap("variable_S")
        extras["audio_sft"] = train_leg(args, "qwen2.5-1.5b", "audio", args.batch, max(5, args.steps // 2), 2, 1, 0, local_rank, False)
        lap("audio_sft")
        extras["audio_sft_blank_biased"] = train_leg(args, "qwen2.5-1.5b", "audio", args.batch, max(5, args.steps // 2), 2, 1, 0,
                                                     local_rank, False, blank_biased=True)
        lap("audio_sft_blank_biased")
        extras["lora_r64"] = train_leg(args, "qwen2.5-1.5b", "text", args.batch, max(5, args.steps // 2), 2, 1, 0, local_rank, False, lora=True)
        lap("lora_r64")
        extras["qwen2.5-7b"] = train_leg(args, "qwen2.5-7b", "text", args.batch, max(5, args.steps // 2), 2, 1, 0, local_rank,
                                          want_decode=not args.no_decode)     # + the 7B decode leg (weight-streaming kernels
                                                                              # for K = 3584 / 18944 since round 5)
        lap("qwen2.5-7b+decode")
        if not args.no_decode:
            try:
                extras["decode_fp32"] = decode_fp32_leg(local_rank, args.batch)
            except Exception as e:                      # (never the loss of the whole line)
                extras["decode_fp32"] = {"value": None, "error": repr(e)[:300]}
            lap("decode_fp32")
            try:
                extras["train_fp32"] = train_fp32_leg(local_rank, args.batch)
            except Exception as e:
                extras["train_fp32"] = {"value": None, "error": repr(e)[:300]}
            lap("train_fp32")
        # the N > 1 step's exchange on hardware with ONE rank: the headline step again with the chunked
        # all-reduce of the gradient bucket through a one-rank RCCL communicator, side stream and event chain included
        try:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                import socket
                with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:      # a port nobody listens on right now
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device(f"cuda:{local_rank}"))
            extras["exchange_1rank"] = train_leg(args, "qwen2.5-1.5b", "text", args.batch, max(5, args.steps // 2), 2, 1, 0, local_rank,
                                                 False, force_exchange=True)
        except Exception as e:                          # (a record of the failure, never the loss of the whole line)
            extras["exchange_1rank"] = {"value": None, "error": repr(e)[:300]}
        finally:
            if dist.is_initialized():
                dist.destroy_process_group()
        lap("exchange_1rank")
    data_path = None
    if world == 1 and headline and not args.no_extra and not args.no_data_path:
        # SURVEY 8f item 1 / VERDICT r4 item 5: the REAL data path next to the synthetic-input figures -- the training entrypoint
        # (ps_slm_amd.finetune_deepspeed.main: jsonl -> wav-in-ark read -> HIP fbank / LFR / CMVN -> collate -> dynamic batching ->
        # reader thread -> step, hipGraph replay) on 2048 generated 30-second utterances in tmpfs; second epoch's rate
        try:
            import shutil
            import tempfile
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_data_path as dp
            base = "/dev/shm" if os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
            croot = tempfile.mkdtemp(prefix="tasu_corpus_", dir=base)
            try:
                tr = dp.make_corpus(os.path.join(croot, "ark"), 2048, "ark")
                t_only = dp.run(os.path.join(croot, "ark"), tr, True, 2, 1)
                t_audio = dp.run(os.path.join(croot, "ark"), tr, False, 2, 1)
            finally:
                shutil.rmtree(croot, ignore_errors=True)
            data_path = {"corpus": "2048 x 30 s, 16-bit wav-in-ark in " + base + ", 16 utterances per batch by the frame budget (4200, ds_rate 5)",
                         "text_only_utterances_per_s": t_only["epoch_utterances_per_s"][-1],
                         "text_only_vs_synthetic": round(t_only["epoch_utterances_per_s"][-1] / main_rec["value"], 3),
                         "audio_wav_utterances_per_s": t_audio["epoch_utterances_per_s"][-1],
                         "audio_wav_vs_synthetic_audio_sft": (round(t_audio["epoch_utterances_per_s"][-1] / extras["audio_sft"]["value"], 3)
                                                              if extras.get("audio_sft") else None),
                         "note": "ps_slm_amd.finetune_deepspeed.main, num_workers_dataloader=1 (one reader thread + 4 decode threads), "
                                 "++use_graphs=true; text_only reads only the audio lengths (the model never looks at the features); "
                                 "audio: random-init encoder, PSD keeps ~476 of 500 frames (S = 628, as audio_sft); FLAC and the "
                                 "in-line loop: profiles/r05_data_path.json (tools/bench_data_path.py)"}
        except Exception as e:                          # (never the loss of the whole line)
            data_path = {"error": repr(e)[:300]}
        lap("data_path")
    if rank == 0:
        line = {"metric": "train utterances/sec (Qwen2.5-1.5B align)" if args.model != "qwen2.5-7b" else "train utterances/sec (Qwen2.5-7B align)",
                "value": main_rec["value"], "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": main_rec["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "bf16", "data": "synthetic"}
        line.update({k: v for k, v in main_rec.items() if k not in ("value", "unit", "ms_per_step")})
        for name, rec in extras.items():
            if SUB_METRICS[name] is not None:
                rec["metric"] = "train utterances/sec" + SUB_METRICS[name]
            line[name] = rec
        if data_path is not None:
            line["data_path"] = data_path
        if world == 1 and not args.no_cpu_baseline and headline:
            line["cpu_baseline"] = cpu_baseline("train1")
            lap("cpu_baseline")
            side = cpu_baselines(["train16", "decode1", "decode16"], 400)
            line["cpu_baselines"] = {"train_B16": side["train16"], "decode_B1": side["decode1"], "decode_B16": side["decode16"]}
            lap("cpu_baselines_B16_decode")
        line["wall_seconds"] = wall                     # where this run's wall clock went (the timed regions are a small part of it)
        emit(line, real_stdout, args.full_record)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
