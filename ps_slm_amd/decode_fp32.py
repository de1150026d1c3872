"""fp32 arithmetic mode of ``slam_model_asr.generate`` (``train_config.use_fp16 = false``).

The reference decodes WITHOUT autocast on fp32 weights (Multitask/inference_batch.py:113-117,146: the model is built in fp32,
``model.eval()``, ``model.generate(**batch)``; Multitask/model/ps-slm.py:660-675 -> HF ``generate(inputs_embeds=..., num_beams=4)``),
so its tokens are those of an fp32 forward pass.  The bf16 path (ps_slm_amd/decode.py) reproduces them only where rounding cannot
matter; this path computes what the reference computes: fp32 projector (LayerNorm -> Linear -> SiLU -> Linear on the fp32 master
weights), fp32 embeddings and residual stream, fp32 q|k|v / RoPE / attention / MLP on fp32 copies of the frozen Qwen2 weights, an
fp32 KV cache, fp32 logits, log-softmax and top-k -- csrc/fp32.hip through the C-ABI (``tasu_f32_*``).  The beam search itself is the
same device-side bookkeeping as the bf16 path's (``tasu_beam_update``, the cache row index, ``DeviceBeam``), and a generated
position is one hipGraph replay.  Nothing is rounded to bf16; sums run in another order than the reference's CPU BLAS (fp32 MFMA,
K ascending, K-range slabs added in ascending order: deterministic).  The audio branch runs the frozen SenseVoice encoder, the CTC
softmax and PSD in fp32 as well (ps_slm_amd/encoder.py: encoder_posterior_fp32; TasuModel.prepare_audio takes it when
``arith == "fp32"`` and no labels are given).  Pinned by exact token equality with the REAL reference on 24 unfiltered random cases,
the 17 rounding-stable ones and the text + audio fixture (tests/test_gpu_model.py).

HBM-bound like the bf16 step, on twice the bytes: 6.2 GB of fp32 weights per generated position at Qwen2.5-1.5B.
"""
import collections

import numpy as np
import torch

from .decode import BEAM_MAX_B, BEAM_MAX_NB, DECODE_GRAPH_CACHE, DONE_POLL_DEPTH, DeviceBeam, effective_min_length
from .model import HD, StepState, rup

F32_MAX_CTX = 2048          # tasu_f32_attn_*: keys per query row


def _gemm_ws(model):
    """The fp32 GEMMs' slab workspace: 16 K-range slabs of the widest narrow projection, or the lm_head's two slabs at 64 beam rows
    (csrc/fp32.hip f32_stream_plan: a problem whose slabs do not fit runs unsplit on the tile kernel)."""
    return model._buf("f32_gemm_ws", (max(16 * 128 * 4096, 2 * 64 * model.geo.llm_vocab),), torch.float32)


def _fragments(model):
    """Fragment-order copies of the matrices the decode step STREAMS (gate|up, down, lm_head: csrc/fp32.hip f32_stream_kernel reads
    1 KiB per wave instruction from them instead of 16 rows x 64 B), made at the first generate(): +5.3 GB at Qwen2.5-1.5B.
    ``TASU_F32_FRAGMENTS=0``: A/B runs on the row-major matrices (the same bits)."""
    import os
    llm = model.llm
    if os.environ.get("TASU_F32_FRAGMENTS", "1") == "0" or not hasattr(model.ops, "f32_to_fragments"):
        return None
    fr = llm.f32.get("frag")
    if fr is None:
        ops, nws = model.ops, _gemm_ws(model).numel()

        def to(w):                                         # only what the streaming kernel will read (Qwen2.5-7B's down projection,
            N, K = w.shape                                 # K = 18944, has no K slice it serves: no copy)
            return ops.f32_to_fragments(w) if ops.lib.tasu_f32_gemm_streams(64, N, K, nws) == 1 else w
        fr = llm.f32["frag"] = dict(layers=[dict(wgu=to(f["wgu"]), wd=to(f["wd"])) for f in llm.f32["layers"]], head=to(llm.f32["head"]))
    return fr


def project_fp32(model, st: StepState):
    """EncoderProjectorLinearSiLU (Multitask/model/projector.py:128-151) in fp32 on the master weights: st.dev['y2_f32'] [Rap, D]."""
    ops, pr = model.ops, model.proj
    if pr.kind != "linear-silu":
        raise NotImplementedError(f"fp32 decode serves the shipped projector (linear-silu), not {pr.kind!r}")
    f32 = torch.float32
    Fap, Rap, K, Kp, Hb, Do = st.Fap, st.Rap, pr.K, pr.Kp, pr.Hb, pr.Do
    if "post" not in st.dev:                                   # text branch: the pseudo-posterior rows (ps-slm.py:337-358)
        post = model._buf("post", (Fap, Kp), f32)
        ops.posterior_build(st.dev["post_ids"], st.dev["post_alpha"], post, Fap, K)
        st.dev["post"] = post
    ws = _gemm_ws(model)
    xn = model._buf("f32_proj_xn", (Fap, Kp), f32)
    ops.layernorm_fwd(st.dev["post"], pr.view(pr.p, "norm.weight"), pr.view(pr.p, "norm.bias"), xn, None, None, Fap, K, model.geo.ln_eps)
    h = model._buf("f32_proj_h", (Rap, Hb), f32)
    ops.f32_gemm(xn, pr.view(pr.p, pr.n_w1), h, Rap, Hb, Kp, bias=pr.view(pr.p, pr.n_b1), act=1, ws=ws)
    y2 = model._buf("f32_proj_y2", (Rap, Do), f32)
    ops.f32_gemm(h, pr.view(pr.p, pr.n_w2), y2, Rap, Do, Hb, bias=pr.view(pr.p, pr.n_b2), ws=ws)
    st.dev["y2_f32"] = y2
    return y2


def _layer_fp32(model, l, x, xn, qkv, ao, gu, act, rows, cos_t, sin_t, attend, ws, cache=None, ctx=0, frag=None):
    """One decoder layer; in: xn = RMSNorm(x, ln1[l]); out: x updated, xn = the NEXT norm of it (ln1[l + 1], or the final norm).
    Every projection carries the row-wise kernel behind it in the launch that sums its K-range slabs (tasu_f32_gemm_qkv_rope,
    _resid_rmsnorm, _swiglu): 9 launches per layer at <= 64 beam rows instead of 13."""
    ops, geo, llm = model.ops, model.geo, model.llm
    D, I, H, G, L = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads, geo.llm_layers
    f, w = llm.f32["layers"][l], llm.layers[l]
    next_norm = llm.layers[l + 1]["ln1"] if l + 1 < L else llm.norm
    kc_l, vc_l, slot = cache if cache is not None else (None, None, None)
    ops.f32_gemm_qkv_rope(xn, f["wqkv"], f["bqkv"], qkv, cos_t, sin_t, rows, H, G, D, ws, kc=kc_l, vc=vc_l, slot=slot, ctx=ctx)
    attend(l, qkv, ao)
    ops.f32_gemm_resid_rmsnorm(ao, f["wo"], x, w["ln2"], xn, rows, D, H * HD, geo.rms_eps, ws, resid=x)
    wgu, wd = (f["wgu"], f["wd"]) if frag is None else (ops.f32_weight(frag["layers"][l]["wgu"], rows, ws), ops.f32_weight(frag["layers"][l]["wd"], rows, ws))
    ops.f32_gemm_swiglu(xn, wgu, gu, act, rows, I, D, ws)
    ops.f32_gemm_resid_rmsnorm(act, wd, x, next_norm, xn, rows, D, I, geo.rms_eps, ws, resid=x)


def _need_f32(model):
    if model.lora is not None:
        raise NotImplementedError("the fp32 path of a LoRA-adapted model is not built (the merged weights exist in bf16 only)")
    if not getattr(model.llm, "f32", None):
        raise RuntimeError("the fp32 path needs the fp32 copies of the LLM weights: build the model with train_config.use_fp16=false "
                           "(model_factory sets LLMWeights.keep_f32 before loading)")


def prompt_pass_fp32(model, st: StepState, on_layer=None):
    """The decoder over the merged prompt in fp32: projector -> embedding merge -> 28 layers (causal attention; a batch's padding is
    on one side: left-padded prompts mask their first ``S - valid`` keys, right-padded training batches need no key mask under the
    causal one -- their padded QUERY rows hold garbage nobody reads).  ``on_layer(l, qkv)``: called with the layer's rotated q|k|v
    (generate() fills its cache there).  Returns (xn0 = the final-normed hidden states [B * S, D], x0 = the residual stream)."""
    ops, geo, llm = model.ops, model.geo, model.llm
    _need_f32(model)
    B, S = st.B, st.S
    if S > F32_MAX_CTX:
        raise ValueError(f"sequence length {S} exceeds the fp32 attention's limit {F32_MAX_CTX}")
    km = np.asarray(st.plan.key_mask)[:, :S].astype(bool)
    valid = km.sum(1).astype(np.int64)
    left = all(km[b, S - valid[b]:].all() for b in range(B))
    right = all(km[b, :valid[b]].all() for b in range(B))
    if not (left or right):
        raise ValueError("the fp32 path expects every row's padding on one side (left: inference collator, right: training collator)")
    M0 = B * S
    D, I, H, G, L = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads, geo.llm_layers
    LDQ = (H + 2 * G) * HD
    scale = HD ** -0.5
    f32 = torch.float32
    buf, d = model._buf, st.dev
    ws = _gemm_ws(model)
    y2 = project_fp32(model, st)
    kstart_b = model._upload("f32_kstart_b", ((S - valid) if left else np.zeros(B, dtype=np.int64)).astype(np.int32))
    x0 = buf("f32_x0", (M0, D), f32)
    ops.f32_embed_merge(llm.embed, y2, d["kind"], d["idx"], x0, M0, D)
    cos0, sin0 = buf("f32_cos0", (M0, HD // 2), f32), buf("f32_sin0", (M0, HD // 2), f32)
    ops.rope_table(d["pos"], cos0, sin0, HD, geo.rope_theta)
    xn0, qkv0, ao0 = buf("f32_xn0", (M0, D), f32), buf("f32_qkv0", (M0, LDQ), f32), buf("f32_ao0", (M0, H * HD), f32)
    gu0, act0 = buf("f32_gu0", (M0, 2 * I), f32), buf("f32_act0", (M0, I), f32)

    def attend_prompt(l, qkv, ao):
        if on_layer is not None:
            on_layer(l, qkv)
        ops.f32_attn_prefill(qkv, kstart_b, ao, B, S, H, G, scale)

    ops.f32_rmsnorm(x0, llm.layers[0]["ln1"], xn0, M0, D, geo.rms_eps)
    for l in range(L):
        _layer_fp32(model, l, x0, xn0, qkv0, ao0, gu0, act0, M0, cos0, sin0, attend_prompt, ws)
    return xn0, x0, valid, left


def forward_fp32(model, st: StepState, compute_loss=True):
    """The eval-mode forward in the reference's fp32 arithmetic (``train_config.use_fp16 = false``: ``evaluation()`` of
    Multitask/utils/deepspeed_utils.py:394-498 and any ``model(**batch)`` outside autocast): fp32 logits for every position
    (``st.dev['logits']`` [B * S, V]), the shifted CE over the labelled rows and the token accuracy (``st.dev['loss_out']`` =
    [mean loss, accuracy, count, 1 / count]).  No activations are kept: there is no fp32 backward."""
    ops, geo, llm = model.ops, model.geo, model.llm
    xn0, _, _, _ = prompt_pass_fp32(model, st)
    M0, D, V = st.B * st.S, geo.llm_dim, geo.llm_vocab
    f32, i32 = torch.float32, torch.int32
    buf, d = model._buf, st.dev
    ws = _gemm_ws(model)
    logits = buf("f32_logits_all", (M0, V), f32)
    ops.f32_gemm(xn0, llm.f32["head"], logits, M0, V, D, ws=ws)
    d.update(logits=logits)
    if not compute_loss:
        return
    row_loss, row_hit = buf("row_loss", (M0,), f32), buf("row_hit", (M0,), i32)
    row_arg, row_lse = buf("row_arg", (M0,), i32), buf("f32_row_lse", (M0,), f32)
    ops.f32_ce(logits, d["shift_labels"], M0, V, row_loss, row_hit, row_arg, row_lse)
    res = buf("loss_out", (4,), f32)
    ops.ce_reduce(row_loss, row_hit, d["shift_labels"], M0, res)
    d.update(loss_out=res, row_arg=row_arg, row_lse=row_lse)


def beam_search_generate_fp32(model, st: StepState, num_beams=4, max_new_tokens=200, min_length=1, length_penalty=1.0,
                              eos_token_id=None, pad_token_id=None):
    """st: a prepared state (prepare_text / prepare_audio).  Returns LongTensor [B, n_new] (CPU)."""
    ops, geo, llm = model.ops, model.geo, model.llm
    _need_f32(model)
    B, S, nb = st.B, st.S, num_beams
    min_length = effective_min_length(min_length, S)
    if not 1 <= nb <= BEAM_MAX_NB:
        raise ValueError(f"num_beams={nb}: the device beam search serves 1..{BEAM_MAX_NB} beams")
    if B > BEAM_MAX_B:
        raise ValueError(f"{B} utterances per generate() call: the device beam search serves at most {BEAM_MAX_B}")
    if max_new_tokens < 1:
        raise ValueError("max_new_tokens must be >= 1")
    ctx = S + max_new_tokens
    if ctx > F32_MAX_CTX:
        raise ValueError(f"prompt {S} + max_new_tokens {max_new_tokens} exceeds the fp32 attention's context limit {F32_MAX_CTX}")
    M0, M, K = B * S, B * nb, 2 * nb
    D, I, H, G, V, L = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads, geo.llm_vocab, geo.llm_layers
    LDQ, W = (H + 2 * G) * HD, G * HD
    eos = geo.eos_id if eos_token_id is None else eos_token_id
    pad = eos if pad_token_id is None else pad_token_id
    scale = HD ** -0.5
    f32, i32 = torch.float32, torch.int32
    buf, d = model._buf, st.dev
    ws = _gemm_ws(model)
    frag = _fragments(model)
    head = llm.f32["head"] if frag is None else ops.f32_weight(frag["head"], M, ws)

    # ---- KV cache (fp32) + the beam row index of the bf16 path
    kc = buf("f32_kc", (L, M * ctx * W), f32)
    vc = buf("f32_vc", (L, M * ctx * W), f32)
    index = buf("dec_index", (M, ctx), i32)
    index_tmp = buf("dec_index_tmp", (M, ctx), i32)
    ops.kv_index_init(index, B, nb, S, ctx)
    ops.kv_index_init(index_tmp, B, nb, S, ctx)
    # ---- prompt pass; every layer's rotated K / V go to the cache row of the utterance's first beam
    xn0, _, valid, left = prompt_pass_fp32(model, st, on_layer=lambda l, qkv_l: ops.f32_kv_fill(qkv_l, kc[l], vc[l], B, S, H, G, nb, ctx))
    if not left:
        raise ValueError("fp32 decode expects left-padded prompts (what the reference's inference collator builds)")
    kstart = model._upload("dec_kstart", np.repeat(S - valid, nb).astype(np.int32), flush=False)
    last_rows = model._upload("dec_last_rows", (np.arange(B, dtype=np.int32) + 1) * S - 1)
    x, xn = buf("f32_x", (M, D), f32), buf("f32_xn", (M, D), f32)
    logits = buf("f32_logits", (M, V), f32)
    ops.embed_rows(xn0, last_rows, xn, B, D)                    # the final-normed last prompt position of every utterance
    ops.f32_gemm(xn, llm.f32["head"], logits, B, V, D, ws=ws)
    tv, ti = buf("dec_topv", (M, K), f32), buf("dec_topi", (M, K), i32)
    bs = DeviceBeam(model, B, nb, max_new_tokens, eos, length_penalty, min_length, S, valid)
    model._last_beam = bs
    topk_ws = buf("f32_topk_ws", (M * 16 * (2 + 2 * K),), f32)               # the row split over 16 workgroups (tasu_f32_logprob_topk)
    ops.f32_logprob_topk(logits, B, V, K, bs.banned, 1, tv, ti, ws=topk_ws)
    ops.beam_update(tv, ti, bs, True)
    qkv, ao = buf("f32_qkv", (M, LDQ), f32), buf("f32_ao", (M, H * HD), f32)
    gu, act = buf("f32_gu", (M, 2 * I), f32), buf("f32_act", (M, I), f32)
    cos, sin = buf("dec_cos", (M, HD // 2), f32), buf("dec_sin", (M, HD // 2), f32)
    kcv, vcv = kc.view(L, M * ctx * W), vc.view(L, M * ctx * W)

    def attend_cache(l, qkv_, ao_):
        ops.f32_attn_decode(qkv_, kcv[l], vcv[l], index, kstart, bs.next_lens, ao_, M, H, G, ctx, scale)

    def device_step():
        """One generated position for all M beams (the launch sequence of decode.py's device_step, fp32 kernels)."""
        ops.kv_index_reorder(index, index_tmp, bs.next_src, bs.next_slot, M, ctx)       # beam reorder of the positions before this one
        ops.kv_index_reorder(index_tmp, index, None, bs.next_slot, M, ctx)
        ops.embed_rows(llm.embed, bs.next_ids, x, M, D)
        ops.rope_table(bs.next_pos, cos, sin, HD, geo.rope_theta)
        ops.f32_rmsnorm(x, llm.layers[0]["ln1"], xn, M, D, geo.rms_eps)
        for l in range(L):
            _layer_fp32(model, l, x, xn, qkv, ao, gu, act, M, cos, sin, attend_cache, ws, cache=(kcv[l], vcv[l], bs.next_slot), ctx=ctx, frag=frag)
        ops.f32_gemm(xn, head, logits, M, V, D, ws=ws)                                 # xn: the final norm, from the last layer's finisher
        ops.f32_logprob_topk(logits, M, V, K, bs.banned, 1, tv, ti, ws=topk_ws)
        ops.beam_update(tv, ti, bs, False)

    use_graphs = model.decode_graphs and model.device.type == "cuda"
    graphs, seen_cnt = model._dec_graphs, model._dec_seen
    key = ("decode_fp32", B, S, nb, ctx, max_new_tokens, int(eos), int(min_length), model._buf_gen)

    def run_step():
        if not use_graphs:
            return device_step()
        for old in [k for k in graphs if k[-1] != model._buf_gen]:
            del graphs[old]
            seen_cnt.pop(old, None)
        g = graphs.get(key)
        if g is not None:
            graphs.move_to_end(key)
            return g.replay()
        seen = seen_cnt.get(key, 0)
        seen_cnt[key] = seen + 1
        if seen < 1:
            return device_step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        import gc
        gc_was = gc.isenabled()
        gc.disable()                                             # (no cyclic collection inside a capture: TasuModel._graphed)
        try:
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                device_step()
        finally:
            if gc_was:
                gc.enable()
        graphs[key] = g
        while len(graphs) > DECODE_GRAPH_CACHE:
            old, _ = graphs.popitem(last=False)
            seen_cnt.pop(old, None)
        g.replay()

    inflight = collections.deque()
    for _ in range(max_new_tokens - 1):
        run_step()
        ev = torch.cuda.Event()
        ev.record()
        inflight.append(ev)
        if len(inflight) > DONE_POLL_DEPTH:
            inflight.popleft().synchronize()
            if int(bs.done_host[0]):
                break
    torch.cuda.synchronize()
    return bs.result(pad)
