"""Decode loop of ``slam_model_asr.generate`` (Multitask/model/ps-slm.py:539-677) on the gfx950 kernels: prefill with
the training forward kernels, then one single-token step per generated position with a per-beam KV cache.  The beam
bookkeeping of HF ``generate(num_beams=4, do_sample=False, early_stopping=False)`` (transformers generation/utils.py
``_beam_search``; dependency not in the reference tree) runs ON THE DEVICE (``tasu_beam_update``: selection among the
rows' 2*nb best log-probs, finished-hypothesis heap, early-stop heuristic, and the next step's token ids / cache rows /
positions), so a generated position is ONE hipGraph replay with no host round trip; the host only polls a pinned
"done" word a couple of steps behind the device and walks the back-pointers once at the end.  ``BeamState`` is the
vectorised host restatement of the same bookkeeping (tests pin it against the oracle's loop, and the kernel against it).

Device per step (M = B*nb rows): beam reorder of the cache's row index -> embedding rows -> 28 x [RMSNorm, qkv GEMM, RoPE,
KV append, cache attention, o GEMM + residual, RMSNorm, gate|up GEMM, SwiGLU, down GEMM + residual] -> RMSNorm -> lm_head
GEMM -> log-softmax top-k -> beam update.  HBM-bound: every step streams the bf16 weights once.
"""
import collections

import numpy as np
import torch

from .model import HD, StepState, rup

NEG = -1.0e9
DECODE_GRAPH_CACHE = 8   # decode-step graphs kept per model (LRU)
BEAM_MAX_NB, BEAM_MAX_B, DECODE_MAX_CTX = 5, 256, 2048   # limits of tasu_beam_update / tasu_decode_step_prologue / tasu_attn_decode


class BeamState:
    """Host restatement of HF's beam-search state update (one call per generated position)."""

    def __init__(self, B, nb, max_new, eos, pad, length_penalty=1.0, min_length=1):
        self.B, self.nb, self.K, self.max_new = B, nb, 2 * nb, max_new
        self.eos, self.lp, self.min_length = eos, length_penalty, min_length
        self.run_seq = np.full((B, nb, max_new), pad, dtype=np.int64)
        self.fin_seq = self.run_seq.copy()
        self.run_scores = np.zeros((B, nb), dtype=np.float32)
        self.run_scores[:, 1:] = NEG
        self.fin_scores = np.full((B, nb), NEG, dtype=np.float32)
        self.fin_len = np.zeros((B, nb), dtype=np.int64)
        self.is_fin = np.zeros((B, nb), dtype=bool)
        self.unsat = np.ones(B, dtype=bool)
        self.cur = 0
        self.done = False

    def ban_eos(self):
        return self.cur < self.min_length

    def update(self, vals, idx):
        """vals/idx: [B, nb, K] per-row top-K log-probs (descending) and token ids.  Returns (tokens [B, nb],
        parent beam [B, nb]) of the beams that keep running."""
        B, nb, K, cur = self.B, self.nb, self.K, self.cur
        acc = vals.astype(np.float32) + self.run_scores[:, :, None]
        flat = acc.reshape(B, nb * K)
        flat_tok = idx.reshape(B, nb * K)
        flat_beam = np.repeat(np.arange(nb), K)[None, :].repeat(B, 0)
        # top-K of the batch's candidates; ties resolved like a flattened [nb * V] top-k (beam, then token id)
        order = np.lexsort((flat_tok, flat_beam, -flat), axis=-1)[:, :K]
        top_lp = np.take_along_axis(flat, order, 1)
        tok = np.take_along_axis(flat_tok, order, 1)
        beam = np.take_along_axis(flat_beam, order, 1)
        cand = np.take_along_axis(self.run_seq, beam[:, :, None], 1).copy()
        cand[:, :, cur] = tok
        stop = (tok == self.eos) | (cur + 1 >= self.max_new)
        run_lp = top_lp + stop.astype(np.float32) * NEG
        nxt = np.argsort(-run_lp, axis=1, kind="stable")[:, :nb]
        self.run_seq = np.take_along_axis(cand, nxt[:, :, None], 1)
        self.run_scores = np.take_along_axis(run_lp, nxt, 1)
        next_tok = np.take_along_axis(tok, nxt, 1)
        next_parent = np.take_along_axis(beam, nxt, 1)
        top_mask = np.arange(K)[None, :] < nb
        just = stop & top_mask
        sc = top_lp / np.float32((cur + 1) ** self.lp)
        sc = sc + (~self.unsat)[:, None].astype(np.float32) * NEG + (~just).astype(np.float32) * NEG
        m_seq = np.concatenate([self.fin_seq, cand], 1)
        m_sc = np.concatenate([self.fin_scores, sc], 1)
        m_len = np.concatenate([self.fin_len, np.full((B, K), cur + 1, dtype=np.int64)], 1)
        m_fin = np.concatenate([self.is_fin, just], 1)
        keep = np.argsort(-m_sc, axis=1, kind="stable")[:, :nb]
        self.fin_seq = np.take_along_axis(m_seq, keep[:, :, None], 1)
        self.fin_scores = np.take_along_axis(m_sc, keep, 1)
        self.fin_len = np.take_along_axis(m_len, keep, 1)
        self.is_fin = np.take_along_axis(m_fin, keep, 1)
        self.cur = cur + 1
        best_run = self.run_scores[:, :1] / np.float32(self.cur ** self.lp)
        worst_fin = np.where(self.is_fin, self.fin_scores.min(1, keepdims=True), np.float32(NEG))
        self.unsat = self.unsat & (best_run > worst_fin).any(-1)
        self.done = not (self.unsat.any() and not stop.all())
        return next_tok, next_parent

    def result(self):
        n = int(self.fin_len[:, 0].max())
        return self.fin_seq[:, 0, :n]


class DeviceBeam:
    """Device-resident beam-search state + the next step's inputs (the arguments of ``tasu_beam_update``)."""

    def __init__(self, model, B, nb, max_new, eos, length_penalty, min_length, S, valid):
        self.B, self.nb, self.max_new, self.eos, self.min_length, self.S = B, nb, max_new, int(eos), int(min_length), S
        up, buf = model._upload, model._buf
        M = B * nb
        i32 = torch.int32
        rs = np.zeros((B, nb), dtype=np.float32)
        rs[:, 1:] = NEG
        self.run_scores = up("bm_run", rs)
        self.fin_scores = up("bm_fin", np.full((B, nb), NEG, dtype=np.float32))
        zeros = np.zeros((B, nb), dtype=np.int32)
        self.fin_len, self.fin_par = up("bm_flen", zeros), up("bm_fpar", zeros)
        self.fin_tok, self.is_fin = up("bm_ftok", zeros), up("bm_isfin", zeros)
        self.unsat = up("bm_unsat", np.ones(B, dtype=np.int32))
        self.bp_tok = buf("bm_bptok", (max_new, B, nb), i32)
        self.bp_par = buf("bm_bppar", (max_new, B, nb), i32)
        # float32(t ** length_penalty) exactly as the host restatement computes it (python float power, then one rounding)
        # (index 0 is never read: a hypothesis has at least one token; 0 ** negative penalty would raise)
        self.len_pow = up("bm_lenpow", np.array([1.0] + [np.float32(float(t) ** float(length_penalty)) for t in range(1, max_new + 2)],
                                                dtype=np.float32))
        self.ctl = up("bm_ctl", np.zeros(2, dtype=np.int32))
        self.valid = up("bm_valid", np.asarray(valid, dtype=np.int32))
        self.next_ids, self.next_src = buf("in_dec_ids", (M,), i32), buf("in_dec_src", (M,), i32)
        self.next_pos, self.next_slot, self.next_lens = buf("in_dec_pos", (M,), i32), buf("in_dec_slot", (M,), i32), buf("in_dec_lens", (M,), i32)
        self.banned = up("dec_banned", np.array([eos if min_length > 0 else -1], dtype=np.int32))
        self.done_host = None
        if model.device.type == "cuda":
            if getattr(model, "_done_host", None) is None:
                model._done_host = torch.zeros(1, dtype=i32).pin_memory()
            self.done_host = model._done_host
            self.done_host.zero_()

    def result(self, pad):
        """Walks the back-pointers of every utterance's best finished hypothesis -> LongTensor [B, n_new] (CPU)."""
        flen = self.fin_len.cpu().numpy()[:, 0]
        fpar, ftok = self.fin_par.cpu().numpy()[:, 0], self.fin_tok.cpu().numpy()[:, 0]
        bpt, bpp = self.bp_tok.cpu().numpy(), self.bp_par.cpu().numpy()
        n = int(flen.max())
        out = np.full((self.B, n), pad, dtype=np.int64)
        for b in range(self.B):
            t = int(flen[b]) - 1
            if t < 0:
                continue
            out[b, t] = ftok[b]
            slot = int(fpar[b])
            for u in range(t - 1, -1, -1):
                out[b, u] = bpt[u, b, slot]
                slot = int(bpp[u, b, slot])
        return torch.from_numpy(out)


DONE_POLL_DEPTH = 2      # generated positions the device may run ahead of the host's look at the "done" word


def effective_min_length(min_length, S):
    """Generated positions during which EOS is banned.  HF counts ``min_length`` INCLUDING the prompt and, when the prompt arrives
    as ``inputs_embeds`` (what the reference passes, Multitask/model/ps-slm.py:660-668), subtracts the embedded prompt's length S
    (padding included) from it: transformers generation/utils.py ``GenerationMixin._prepare_generated_length``,
    ``min_length = max(min_length - inputs_tensor.shape[1], 0)``.  With the reference's default ``min_length=1`` EOS is never
    banned.  (Rounds 1-5 banned the first ``min_length`` generated positions; found in round 6 on an unfiltered decode case whose
    reference output is an immediate EOS: tests/golden/mid_generate_fp32.npz.)"""
    return max(int(min_length) - int(S), 0)


def beam_search_generate(model, st: StepState, num_beams=4, max_new_tokens=200, min_length=1, length_penalty=1.0,
                         eos_token_id=None, pad_token_id=None):
    """st: a prepared state whose projector output (st.dev['y2']) is ready.  Returns LongTensor [B, n_new] (CPU)."""
    if model.lora is not None and model._lora_run is not None:
        # use_peft: prefill and the decode loop run on the merged weights W + s B A (ps_slm_amd/lora.py: merged_llm)
        from .lora import merged_llm
        keep = (model.llm, model._lora_run)
        model.llm, model._lora_run = merged_llm(model), None
        try:
            return beam_search_generate(model, st, num_beams, max_new_tokens, min_length, length_penalty, eos_token_id, pad_token_id)
        finally:
            model.llm, model._lora_run = keep
    ops, geo, llm = model.ops, model.geo, model.llm
    B, S, nb = st.B, st.S, num_beams
    min_length = effective_min_length(min_length, S)
    # limits of the device beam search (tasu_beam_update, tasu_decode_step_prologue: include/tasu_hip.h), checked BEFORE the prefill
    if not 1 <= nb <= BEAM_MAX_NB:
        raise ValueError(f"num_beams={nb}: the device beam search serves 1..{BEAM_MAX_NB} beams")
    if B > BEAM_MAX_B:
        raise ValueError(f"{B} utterances per generate() call: the device beam search serves at most {BEAM_MAX_B}")
    if S + max_new_tokens > DECODE_MAX_CTX:
        raise ValueError(f"prompt {S} + max_new_tokens {max_new_tokens} exceeds the cache attention's context limit {DECODE_MAX_CTX}")
    if max_new_tokens < 1:
        raise ValueError("max_new_tokens must be >= 1")
    M, K = B * nb, 2 * nb
    D, I, H, G, V, L = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads, geo.llm_vocab, geo.llm_layers
    Vp, LDQ, W = rup(V, 64), (H + 2 * G) * HD, G * HD
    eos = geo.eos_id if eos_token_id is None else eos_token_id
    pad = eos if pad_token_id is None else pad_token_id
    ctx = S + max_new_tokens
    scale = HD ** -0.5
    bf, f32, i32 = torch.bfloat16, torch.float32, torch.int32
    buf = model._buf
    # ---- prefill with the training-forward kernels (no loss)
    model.forward_llm(st, compute_loss=False, need_backward=False, logits_rows="none")
    # decode-step weights in the order the streaming kernels consume them (built once per model), and the decision whether the
    # step's bf16 activations travel between its kernels in that order too (ops.begin_decode)
    llm.prepare_decode(ops)
    ops.begin_decode(D, H * HD, I)
    try:
        return _decode_after_prefill(model, st, nb, max_new_tokens, min_length, length_penalty, eos, pad)
    finally:
        ops.end_decode()


def layers_per_gemm(ops, geo, layers, final_norm, x, x2, xn, qkv, ao, act, cos, sin, kc, vc, index, kstart, slot, lens, M, ctx, ws,
                    normed=False):
    """One generated position through the decoder layers with one launch per GEMM (6-7 launches per layer): the path for more
    than 64 beam rows as well (64-row chunks).
    In: x [M, D] fp32 (token embeddings; ``normed``: xn already holds layer 0's input norm of it); out: xn = the final-normed
    hidden state; K/V appended at ``slot``."""
    D, I, H, G, L = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads, len(layers)
    W = G * HD
    scale = HD ** -0.5
    # the weight-streaming kernels take at most 64 rows: more beams than that (B > 16 at 4 beams) run them in row chunks
    # (the weights are then streamed once per chunk; K/V, attention and top-k are not chunked)
    chunks = [(m0, min(64, M - m0)) for m0 in range(0, M, 64)]
    kcv, vcv = kc.view(L, M, ctx * W), vc.view(L, M, ctx * W)
    prenorm = bool(getattr(ops, "prenorm_ok", None) and ops.prenorm_ok(D, H * HD, I))
    prenorm_in = bool(prenorm and ops.prenorm_in_ok(D, I) and len(chunks) <= 8)
    in_ssq = [None] * len(chunks)                                               # per chunk: the input norm's partial sums of squares
    if not normed:                                                               # (the step prologue has done it already)
        for m0, mc in chunks:
            ops.dec_rmsnorm(x[m0:m0 + mc], layers[0]["ln1"], xn[m0:m0 + mc], geo.rms_eps)
    for l, w in enumerate(layers):
        next_norm = layers[l + 1]["ln1"] if l + 1 < L else final_norm            # the norm that consumes this layer's output
        for ci, (m0, mc) in enumerate(chunks):                                   # qkv projection + bias + RoPE + cache append
            r = slice(m0, m0 + mc)
            if in_ssq[ci] is not None:        # xn = bf16(ln1 . x) from the previous layer's slab finish: rstd on the accumulators
                ops.gemm_skinny_qkv_rope(xn[r], w["wqkv"], w["bqkv"], qkv[r], mc, H, G, D, cos[r], sin[r], kcv[l, r], vcv[l, r],
                                         slot[r], ctx, ws, sumsq=in_ssq[ci], eps=geo.rms_eps)
            else:
                ops.gemm_skinny_qkv_rope(xn[r], w["wqkv"], w["bqkv"], qkv[r], mc, H, G, D, cos[r], sin[r], kcv[l, r], vcv[l, r],
                                         slot[r], ctx, ws)
        ops.attn_decode(qkv, kc[l], vc[l], index, kstart, lens, ao, M, H, G, ctx, scale)
        for ci, (m0, mc) in enumerate(chunks):                                   # projections with residual + next norm fused
            r = slice(m0, m0 + mc)
            if prenorm:
                # o projection + residual; the post-attention norm travels inside it and gate|up (no launch of its own)
                ssq = ops.gemm_skinny_prenorm(ao[r], w["wo"], x2[r], x[r], mc, D, H * HD, w["ln2"], xn[r])
                ops.gemm_skinny_swiglu(xn[r], w["wgu"], act[r], mc, I, D, ws, sumsq=ssq, eps=geo.rms_eps)
            else:
                ops.gemm_skinny_norm(ao[r], w["wo"], x2[r], x[r], mc, D, H * HD, w["ln2"], xn[r], geo.rms_eps, ws)
                ops.gemm_skinny_swiglu(xn[r], w["wgu"], act[r], mc, I, D, ws)
            if prenorm_in and l + 1 < L:      # the next layer's input norm travels with the slab finish and that layer's q|k|v
                in_ssq[ci] = ops.gemm_skinny_norm(act[r], w["wd"], x[r], x2[r], mc, D, I, next_norm, xn[r], geo.rms_eps, ws, prenorm_slot=ci)
            else:
                ops.gemm_skinny_norm(act[r], w["wd"], x[r], x2[r], mc, D, I, next_norm, xn[r], geo.rms_eps, ws)
                in_ssq[ci] = None


def _decode_after_prefill(model, st, nb, max_new_tokens, min_length, length_penalty, eos, pad):
    ops, geo, llm = model.ops, model.geo, model.llm
    B, S = st.B, st.S
    M, K = B * nb, 2 * nb
    Mp = rup(M, 64)                                                         # fragment-order buffers hold whole 64-row chunks
    D, I, H, G, V, L = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads, geo.llm_vocab, geo.llm_layers
    Vp, LDQ, W = rup(V, 64), (H + 2 * G) * HD, G * HD
    ctx = S + max_new_tokens
    scale = HD ** -0.5
    bf, f32, i32 = torch.bfloat16, torch.float32, torch.int32
    buf = model._buf
    d = st.dev
    # KV cache [L][M, ctx, W] + the beam row index (include/tasu_hip.h): the prompt is stored once per utterance and a
    # beam reorder permutes 4-byte index entries once per step instead of copying K/V in every layer
    kc = buf("dec_kc", (L, M * ctx * W), bf)
    vc = buf("dec_vc", (L, M * ctx * W), bf)
    index = buf("dec_index", (M, ctx), i32)
    index_tmp = buf("dec_index_tmp", (M, ctx), i32)
    ops.kv_index_init(index, B, nb, S, ctx)
    ops.kv_index_init(index_tmp, B, nb, S, ctx)
    for l in range(L):
        ops.kv_fill(d["qkv"][l], kc[l], vc[l], B, S, H, G, nb, ctx)
    valid = st.plan.key_mask[:, :S].sum(1).astype(np.int64)               # real tokens per prompt
    kstart_h = np.repeat(S - valid, nb).astype(np.int32)                   # left padding is at the front
    kstart = model._upload("dec_kstart", kstart_h)
    # logits of the last prompt position of every batch row
    last_rows = model._upload("dec_last_rows", (np.arange(B, dtype=np.int32) + 1) * S - 1)
    xl = buf("dec_xlast", (B, D), f32)
    ops.embed_rows(d["xs"][2 * L], last_rows, xl, B, D)                    # row gather out of the residual stream
    xn = buf("dec_xn", (Mp, D), bf)
    logits = buf("dec_logits", (M, Vp), bf)
    ops.rmsnorm_fwd(xl, llm.norm, xn[:B], None, geo.rms_eps)
    ops.gemm(xn, llm.head, logits, B, V, D)
    tv = buf("dec_topv", (M, K), f32)
    ti = buf("dec_topi", (M, K), i32)
    bs = DeviceBeam(model, B, nb, max_new_tokens, eos, length_penalty, min_length, S, valid)
    model._last_beam = bs                                                  # (tests read the final scores / back-pointers)
    ops.logprob_topk(logits, B, V, K, bs.banned, 1, tv, ti)
    ops.beam_update(tv, ti, bs, True)                                      # first position: only beam 0 exists
    x = buf("dec_x", (M, D), f32)
    x2 = buf("dec_x2", (M, D), f32)
    qkv = buf("dec_qkv", (M, LDQ), bf)
    ao = buf("dec_ao", (Mp, H * HD), bf)
    act = buf("dec_act", (Mp, I), bf)
    cos = buf("dec_cos", (M, HD // 2), f32)
    sin = buf("dec_sin", (M, HD // 2), f32)
    ws = buf("dec_gemm_ws", (32 * 64 * rup(max(V, 2 * I), 96),), f32)          # tasu_gemm_skinny_bf16 split-K slabs
    # per-step device inputs live in fixed buffers (written by tasu_beam_update) so that the step replays as a hipGraph
    ids_d, pos_d, slot_d, lens_d, src_d = bs.next_ids, bs.next_pos, bs.next_slot, bs.next_lens, bs.next_src

    # The weight-streaming kernels take at most 64 rows: more beams than that (B > 16 at 4 beams) run them in row chunks
    # (the weights are then streamed once per chunk; K/V, attention and top-k are not chunked).
    chunks = [(m0, min(64, M - m0)) for m0 in range(0, M, 64)]
    kcv, vcv = kc.view(L, M, ctx * W), vc.view(L, M, ctx * W)
    def device_step():
        """One generated position for all M beams: beam reorder of the row index (parents of the previous step), then the
        28-layer single-token pass over the cache, lm_head, the per-row top-k and the beam update."""
        ops.decode_step_prologue(llm.embed, ids_d, x, llm.layers[0]["ln1"], xn, geo.rms_eps, pos_d, cos, sin, HD, geo.rope_theta, index,
                                 index_tmp, src_d, slot_d, nb, M, D, ctx)
        per_gemm_layers()
        for m0, mc in chunks:
            ops.gemm_skinny(xn[m0:m0 + mc], llm.head, logits[m0:m0 + mc], mc, V, D, ws)
        ops.logprob_topk(logits, M, V, K, bs.banned, 1, tv, ti)
        ops.beam_update(tv, ti, bs, False)

    def per_gemm_layers():
        layers_per_gemm(ops, geo, llm.layers, llm.norm, x, x2, xn, qkv, ao, act, cos, sin, kc, vc, index, kstart, slot_d, lens_d,
                        M, ctx, ws, normed=True)

    # hipGraph replay of device_step (~430 launches): the first step of a shape runs eagerly, the second is captured.
    # A graph is only valid for the buffers it was captured on (grow-only workspace: same generation = same addresses) and
    # for the scalars baked into its kernel arguments; a small LRU bounds the cache (real data gives almost every batch its
    # own prompt length).
    use_graphs = model.decode_graphs and model.device.type == "cuda"
    graphs, seen_cnt = model._dec_graphs, model._dec_seen
    # the launch / layout switches decide which kernels device_step issues: a graph captured under one setting must not be
    # replayed under another (in-process A/B runs)
    switches = tuple(bool(getattr(ops, n, False)) for n in ("use_stream", "dec_frag", "dec_frag_act", "dec_down_slabs", "dec_prologue"))
    key = ("decode", B, S, nb, ctx, max_new_tokens, int(eos), int(min_length), switches, model._buf_gen)

    def run_step():
        if not use_graphs:
            return device_step()
        for old in [k for k in graphs if k[-1] != model._buf_gen]:
            del graphs[old]                                      # captured on addresses that have since been freed
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

    if model.device.type == "cuda":
        # the host issues positions ahead of the device and looks at the pinned "done" word DONE_POLL_DEPTH positions late
        # (an event per position); positions issued after the device finished are no-ops for the beam state
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
    else:
        while not int(bs.ctl[1]):
            run_step()
    return bs.result(pad)
