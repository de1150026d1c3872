"""Frozen SenseVoiceSmall front end of the audio path on the gfx950 kernels:
query-token prepend (Multitask/model/ps-slm.py:430-443) -> SANM encoder (Multitask/model/SenseVoice.py:548-579:
x*sqrt(d)+sinusoidal PE, 1 + 49 layers, after_norm, 20 tp layers, tp_norm) -> CTC head + softmax (ps-slm.py:450-454)
-> PSD (ps-slm.py:237-317).  Forward only (the encoder is frozen and its input needs no gradient).

Per layer (SenseVoice.py:324-400, pre-norm, eval): LayerNorm fp32 -> fused q|k|v GEMM (bf16, K padded 560->576 for
layer 0) -> bidirectional 4-head attention with key padding (tasu_attn_fwd, causal=0) -> linear_out GEMM fused with the residual add -> + FSMN memory (depthwise conv over time on
masked v, tasu_fsmn_fwd) -> LayerNorm -> FFN GEMM / ReLU / GEMM fused with the residual add.
"""
import math

import numpy as np
import torch

from .ops import GEMM_BF16, GEMM_RESID

HD = 128


def rup(x, m):
    return (x + m - 1) // m * m


class EncoderWeights:
    def __init__(self, geo, device, keep_f32=False):
        self.geo, self.device = geo, torch.device(device)
        self.keep_f32 = bool(keep_f32)        # fp32 copies next to the bf16 ones (fp32 arithmetic mode of generate(): encoder_posterior_fp32)
        if geo.enc_dim // geo.enc_heads != HD:
            raise ValueError("the gfx950 attention kernel needs d_k = 128 (SenseVoiceSmall: 512 / 4)")
        self.layers = []
        self.embed = None          # fp32 [16, F]
        self.after_norm = self.tp_norm = None
        self.ctc_w = self.ctc_b = None

    def _lin(self, w, b, kpad=None):
        dev = self.device
        w = w.to(dev, torch.float32)
        if kpad is not None and kpad != w.shape[1]:
            wp = torch.zeros(w.shape[0], kpad, device=dev)
            wp[:, : w.shape[1]] = w
            w = wp
        if self.keep_f32:
            self._f32_last = (w.contiguous(), b.to(dev, torch.float32).contiguous())
        return w.to(torch.bfloat16).contiguous(), b.to(dev, torch.bfloat16).contiguous()

    def _add_layer(self, sd, p, in_dim):
        dev, f32 = self.device, torch.float32
        kp = rup(in_dim, 64)
        f32w = {}
        def lin(tag, w, b, kpad=None):
            out = self._lin(w, b, kpad)
            if self.keep_f32:
                f32w["w" + tag], f32w["b" + tag] = self._f32_last
            return out
        wqkv, bqkv = lin("qkv", sd[p + "self_attn.linear_q_k_v.weight"], sd[p + "self_attn.linear_q_k_v.bias"], kp)
        wout, bout = lin("out", sd[p + "self_attn.linear_out.weight"], sd[p + "self_attn.linear_out.bias"])
        w1, b1 = lin("1", sd[p + "feed_forward.w_1.weight"], sd[p + "feed_forward.w_1.bias"])
        w2, b2 = lin("2", sd[p + "feed_forward.w_2.weight"], sd[p + "feed_forward.w_2.bias"])

        def ln(name, n):
            g = torch.zeros(rup(n, 64), device=dev)
            b = torch.zeros(rup(n, 64), device=dev)
            g[:n] = sd[p + name + ".weight"].to(dev, f32)
            b[:n] = sd[p + name + ".bias"].to(dev, f32)
            return g, b

        E, ks = self.geo.enc_dim, self.geo.enc_kernel
        self.layers.append(dict(in_dim=in_dim, kp=kp, n1=ln("norm1", in_dim), n2=ln("norm2", E), wqkv=wqkv, bqkv=bqkv,
                                wout=wout, bout=bout, w1=w1, b1=b1, w2=w2, b2=b2,
                                fsmn=sd[p + "self_attn.fsmn_block.weight"].to(dev, f32).reshape(E, ks).contiguous(), f32=f32w))

    def load_reference_state_dict(self, sd, pre="encoder."):
        geo, dev, f32 = self.geo, self.device, torch.float32
        self.layers = []
        self._add_layer(sd, pre + "encoder.encoders0.0.", geo.feat_dim)
        for i in range(geo.enc_blocks - 1):
            self._add_layer(sd, f"{pre}encoder.encoders.{i}.", geo.enc_dim)
        self.n_main = len(self.layers)
        for i in range(geo.enc_tp_blocks):
            self._add_layer(sd, f"{pre}encoder.tp_encoders.{i}.", geo.enc_dim)
        self.embed = sd[pre + "embed.weight"].to(dev, f32).contiguous()
        self.after_norm = (sd[pre + "encoder.after_norm.weight"].to(dev, f32), sd[pre + "encoder.after_norm.bias"].to(dev, f32))
        self.tp_norm = (sd[pre + "encoder.tp_norm.weight"].to(dev, f32), sd[pre + "encoder.tp_norm.bias"].to(dev, f32))
        self.ctc_w, self.ctc_b = self._lin(sd[pre + "ctc.ctc_lo.weight"], sd[pre + "ctc.ctc_lo.bias"])
        self.ctc_f32 = self._f32_last if self.keep_f32 else None

    def init_random(self, seed):
        from .synthetic import random_state_dict
        sd = {k: v for k, v in random_state_dict(self.geo, seed, with_encoder=True).items() if k.startswith("encoder.")}
        self.load_reference_state_dict(sd)


QUERY_ROWS = (0, 1, 2, 2)   # language, event, emotion, textnorm query ids (ps-slm.py:430-442)


def encoder_posterior(model, feats, feat_lens, want_post=True):
    """feats [B, T, F] float (host or device), feat_lens [B].  Returns (post fp32 [B*Te, Kp] device, Te, lens int32
    device [B]) where row b*Te + 4 + t is frame t of utterance b (first 4 rows = query tokens).  ``want_post=False`` (the
    training / decode step): the CTC head's bf16 LOGITS [B*Te, Kp] are returned instead -- PSD works from them
    (psd_on_device(logits=True)) and the fp32 posterior of all frames (808 MB per 16 x 504 frames) is never written."""
    ops, geo, enc, dev = model.ops, model.geo, model.encoder, model.device
    if enc is None:
        raise RuntimeError("the audio path needs encoder weights (model_factory(..., with_encoder=True) or encoder_path)")
    B, T, Fd = feats.shape
    E, Hh, Ff, V = geo.enc_dim, geo.enc_heads, geo.enc_ffn, geo.ctc_vocab
    Te, Kp = T + 4, rup(V, 64)
    M = B * Te
    Spad = rup(Te, 64)
    f32, bf = torch.float32, torch.bfloat16
    buf = model._buf
    # [query rows | features]: host-side concat is plumbing (18 MB H2D per 16 utterances, like the reference)
    x0 = buf("enc_x0", (B, Te, Fd), f32)
    x0[:, :4].copy_(enc.embed[list(QUERY_ROWS)].unsqueeze(0).expand(B, -1, -1))
    x0[:, 4:].copy_(feats.to(dev, f32, non_blocking=True))
    lens_h = (np.asarray(feat_lens.cpu() if isinstance(feat_lens, torch.Tensor) else feat_lens).astype(np.int64) + 4)
    lens = model._upload("enc_lens", lens_h.astype(np.int32))
    km = np.zeros((B, Spad), dtype=np.uint8)
    for b in range(B):
        km[b, : int(lens_h[b])] = 1
    key_mask = model._upload("enc_key_mask", km)
    # everything below is a fixed launch sequence for a given (B, T): ~12 launches x 70 blocks, replayed as one hipGraph
    # when the model runs with graphs (the uploads above stay outside the captured region)
    out = {}
    with ops.alt_workspace("encoder"):     # its own split-K workspace: the pass may run on a side stream (TasuModel.prefetch_encoder)
        model.graphed_region(("encoder", B, T, bool(want_post)),
                             lambda: out.update(post=_encoder_body(model, x0, lens, key_mask, B, T, want_post)))
    if not want_post:
        return model._buf("enc_ctc_logits", (M, Kp), bf), Te, lens
    return model._buf("enc_post", (M, Kp), f32), Te, lens


def _encoder_body(model, x0, lens, key_mask, B, T, want_post=True):
    ops, geo, enc = model.ops, model.geo, model.encoder
    Fd = geo.feat_dim
    E, Hh, Ff, V = geo.enc_dim, geo.enc_heads, geo.enc_ffn, geo.ctc_vocab
    Te, Kp = T + 4, rup(V, 64)
    M = B * Te
    Spad = rup(Te, 64)
    f32, bf = torch.float32, torch.bfloat16
    buf = model._buf
    x = buf("enc_x", (M, Fd), f32)
    ops.sinusoid_pe(x0.view(M, Fd), x, B, Te, Fd, float(E) ** 0.5)
    zero_res = buf("enc_zero", (M, E), f32)
    zero_res.zero_()
    qkv = buf("enc_qkv", (M, 3 * E), bf)
    ao = buf("enc_ao", (M, E), bf)
    lse = buf("enc_lse", (B * Hh * Spad,), f32)
    xa = buf("enc_xa", (M, E), f32)
    xb = buf("enc_xb", (M, E), f32)
    h = buf("enc_h", (M, Ff), bf)
    scale = HD ** -0.5
    cur = x                                     # layer input (fp32), width in_dim
    for li, w in enumerate(enc.layers):
        if li == enc.n_main:                    # after_norm between the main and the tp stacks (SenseVoice.py:569)
            nxt = xa if cur is not xa else xb
            ops.layernorm_fwd(cur, enc.after_norm[0], enc.after_norm[1], nxt, None, None, M, E, 1e-5)
            cur = nxt
        xn = buf("enc_xn", (M, w["kp"]), bf)
        ops.layernorm_fwd(cur, w["n1"][0], w["n1"][1], xn, None, None, M, w["in_dim"], 1e-5)
        ops.gemm(xn, w["wqkv"], qkv, M, 3 * E, w["kp"], bias=w["bqkv"])
        ops.attn_fwd(qkv, None, key_mask, ao, lse, B, Te, Hh, Hh, scale, False)
        mid = xa if cur is not xa else xb
        resid = cur if w["in_dim"] == E else zero_res                           # no residual on layer 0 (:372-389)
        ops.gemm(ao, w["wout"], mid, M, E, E, bias=w["bout"], resid=resid, mode=GEMM_RESID)
        xn2 = buf("enc_xn2", (M, E), bf)
        ops.fsmn_ln_fwd(qkv[:, 2 * E:], 3 * E, w["fsmn"], lens, mid, w["n2"][0], w["n2"][1], xn2, B, Te, E, geo.enc_kernel, 1e-5)   # x += fsmn(v); norm2
        ops.gemm_bias_relu(xn2, w["w1"], h, M, Ff, E, w["b1"])                  # w_1 + ReLU: the ReLU in the GEMM's epilogue
        out = xb if mid is xa else xa
        ops.gemm(h, w["w2"], out, M, E, Ff, bias=w["b2"], resid=mid, mode=GEMM_RESID)
        cur = out
    if len(enc.layers) == enc.n_main:           # no tp layers: after_norm still applies
        nxt = xa if cur is not xa else xb
        ops.layernorm_fwd(cur, enc.after_norm[0], enc.after_norm[1], nxt, None, None, M, E, 1e-5)
        cur = nxt
    encb = buf("enc_outb", (M, E), bf)
    if getattr(model, "raw_features", False):
        # the raw-feature branch keeps the encoder's output states (fp32, as LayerNorm leaves them under autocast) for PSD
        encf = buf("enc_outf", (M, E), f32)
        ops.layernorm_fwd(cur, enc.tp_norm[0], enc.tp_norm[1], encf, None, None, M, E, 1e-5)
        ops.cast_bf16(encf, encb)
    else:
        ops.layernorm_fwd(cur, enc.tp_norm[0], enc.tp_norm[1], encb, None, None, M, E, 1e-5)     # tp_norm, bf16 for the CTC GEMM
    logits = buf("enc_ctc_logits", (M, Kp), bf)
    ops.gemm(encb, enc.ctc_w, logits, M, V, E, bias=enc.ctc_b)
    if not want_post:
        return logits
    post = buf("enc_post", (M, Kp), f32)
    ops.softmax_rows(logits, post, M, V)
    return post


def encoder_posterior_fp32(model, feats, feat_lens):
    """The frozen encoder + CTC head + softmax in fp32 (train_config.use_fp16 = false: the reference's inference runs
    SenseVoiceEncoderSmall.forward without autocast, Multitask/model/ps-slm.py:430-454 under inference_batch.py:113-117): fp32
    LayerNorms, fp32 GEMMs (tasu_f32_gemm_nt on fp32 copies of the weights), fp32 bidirectional attention and FSMN (csrc/fp32.hip).
    Returns (posterior fp32 [B * Te, Kp], Te, lens int32 device [B]) like encoder_posterior(want_post=True)."""
    ops, geo, enc, dev = model.ops, model.geo, model.encoder, model.device
    if enc is None or not enc.keep_f32:
        raise RuntimeError("fp32 audio decode needs the encoder's fp32 weight copies (train_config.use_fp16=false before loading)")
    B, T, Fd = feats.shape
    E, Hh, Ff, V = geo.enc_dim, geo.enc_heads, geo.enc_ffn, geo.ctc_vocab
    Te, Kp = T + 4, rup(V, 64)
    M = B * Te
    f32 = torch.float32
    buf = model._buf
    x0 = buf("enc_x0", (B, Te, Fd), f32)
    x0[:, :4].copy_(enc.embed[list(QUERY_ROWS)].unsqueeze(0).expand(B, -1, -1))
    x0[:, 4:].copy_(feats.to(dev, f32, non_blocking=True))
    lens_h = (np.asarray(feat_lens.cpu() if isinstance(feat_lens, torch.Tensor) else feat_lens).astype(np.int64) + 4)
    lens = model._upload("enc_lens", lens_h.astype(np.int32))
    ws = buf("f32_gemm_ws", (16 * 128 * 4096,), f32)
    x = buf("enc_x", (M, Fd), f32)
    ops.sinusoid_pe(x0.view(M, Fd), x, B, Te, Fd, float(E) ** 0.5)
    qkv = buf("f32_enc_qkv", (M, 3 * E), f32)
    ao = buf("f32_enc_ao", (M, E), f32)
    xa, xb = buf("enc_xa", (M, E), f32), buf("enc_xb", (M, E), f32)
    h = buf("f32_enc_h", (M, Ff), f32)
    scale = HD ** -0.5
    cur = x
    for li, w in enumerate(enc.layers):
        f = w["f32"]
        if li == enc.n_main:
            nxt = xa if cur is not xa else xb
            ops.layernorm_fwd(cur, enc.after_norm[0], enc.after_norm[1], nxt, None, None, M, E, 1e-5)
            cur = nxt
        xn = buf("f32_enc_xn", (M, w["kp"]), f32)
        ops.layernorm_fwd(cur, w["n1"][0], w["n1"][1], xn, None, None, M, w["in_dim"], 1e-5)
        ops.f32_gemm(xn, f["wqkv"], qkv, M, 3 * E, w["kp"], bias=f["bqkv"], ws=ws)
        ops.f32_attn_prefill(qkv, None, ao, B, Te, Hh, Hh, scale, klen=lens)
        mid = xa if cur is not xa else xb
        ops.f32_gemm(ao, f["wout"], mid, M, E, E, bias=f["bout"], resid=cur if w["in_dim"] == E else None, ws=ws)   # no residual on layer 0
        ops.f32_fsmn(qkv[:, 2 * E:], 3 * E, w["fsmn"], lens, mid, B, Te, E, geo.enc_kernel)                      # x += fsmn(v)
        xn2 = buf("f32_enc_xn2", (M, E), f32)
        ops.layernorm_fwd(mid, w["n2"][0], w["n2"][1], xn2, None, None, M, E, 1e-5)
        ops.f32_gemm(xn2, f["w1"], h, M, Ff, E, bias=f["b1"], act=2, ws=ws)
        out = xb if mid is xa else xa
        ops.f32_gemm(h, f["w2"], out, M, E, Ff, bias=f["b2"], resid=mid, ws=ws)
        cur = out
    if len(enc.layers) == enc.n_main:
        nxt = xa if cur is not xa else xb
        ops.layernorm_fwd(cur, enc.after_norm[0], enc.after_norm[1], nxt, None, None, M, E, 1e-5)
        cur = nxt
    encf = buf("enc_outf", (M, E), f32)
    ops.layernorm_fwd(cur, enc.tp_norm[0], enc.tp_norm[1], encf, None, None, M, E, 1e-5)
    logits = buf("f32_enc_logits", (M, Kp), f32)
    if Kp > V:
        logits[:, V:].zero_()
    ops.f32_gemm(encf, enc.ctc_f32[0], logits, M, V, E, bias=enc.ctc_f32[1], ws=ws)
    post = buf("enc_post", (M, Kp), f32)
    ops.softmax_rows(logits, post, M, V)
    return post, Te, lens


def psd_on_device(model, post, B, T, Te, feat_lens_dev, do_psd=True, k=1, feats=None, logits=False):
    """PSD over frames 4.. of every utterance.  The decisions (run merging, blank filter) come from the posterior; the rows that
    are kept / averaged are the posterior's (``feats`` None) or those of ``feats`` (fp32 [B * Te, width]: the raw-feature
    branch).  ``k``: frames per projector row -- the batch tensor's trailing Lmax % k frames are dropped (projector.py:41-45).
    ``logits``: ``post`` holds the CTC head's bf16 logits; per-frame argmax / blank probability / softmax statistics come from
    them and the posterior is evaluated for the kept frames only.
    Returns (rows fp32 [rup(B * Lmax, 64 k), width padded to 64], new_lens host int64 [B] (untruncated), Lmax)."""
    ops, geo = model.ops, model.geo
    V = geo.ctc_vocab
    buf = model._buf
    body = post[4:]                                  # row (b, t) = b*Te + t of this view
    fid = buf("psd_fid", (B * T,), torch.int32)
    fbl = buf("psd_fbl", (B * T,), torch.float32)
    ss = buf("psd_ss", (B * T,), torch.int32)
    sl = buf("psd_sl", (B * T,), torch.int32)
    nl = buf("psd_nl", (B,), torch.int32)
    if logits:
        fst = buf("psd_fstat", (B * T, 2), torch.float32)
        ops.psd_logit_stats(body, feat_lens_dev, fid, fbl, fst, B, T, Te, V, geo.blank_id)
    else:
        ops.psd_frame_stats(body, feat_lens_dev, fid, fbl, B, T, Te, V, geo.blank_id)
    thr = 0.90 if do_psd else 2.0                    # do_psd=false keeps every frame (ps-slm.py:472-473)
    blank = geo.blank_id if do_psd else -2           # ... and merges nothing
    ops.psd_plan(fid, fbl, feat_lens_dev, ss, sl, nl, B, T, blank, thr)
    new_lens = nl.cpu().numpy().astype(np.int64)     # one small D2H sync per batch (the reference syncs per frame)
    Lmax = (int(new_lens.max()) // k) * k if B else 0
    if Lmax == 0:
        raise ValueError("PSD removed every frame of every utterance (all-blank batch)" if k == 1 else
                         f"PSD left fewer than {k} frames in every utterance: no projector row")
    src, W = (body, V) if feats is None else (feats[4:], feats.shape[1])
    Wp = rup(W, 64)
    Fap = rup(B * Lmax, 64 * k)
    rows = buf("post", (Fap, Wp), torch.float32)
    if Fap > B * Lmax:
        rows[B * Lmax:].zero_()
    if logits and feats is None:
        ops.psd_gather_softmax(body, fst, ss, sl, nl, rows, B, T, Te, Lmax, V)
    else:
        ops.psd_gather(src, ss, sl, nl, rows, B, T, Te, Lmax, W)
    return rows, new_lens, Lmax
