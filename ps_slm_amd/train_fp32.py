"""fp32 arithmetic mode of the TRAINING step (``train_config.use_fp16 = false``: the reference's shipped recipe,
Multitask/scripts/finetune_deespeed_sensevoice.sh:37 -- forward and backward outside autocast, Multitask/utils/deepspeed_utils.py:160,
205-236).  Same network, same schedule as ps_slm_amd/model.py's bf16 step, every tensor fp32:

  forward   posterior -> LayerNorm -> Linear -> SiLU -> Linear (fp32 masters) -> embedding merge -> 28 x [RMSNorm, q|k|v + bias + RoPE,
            causal attention, o + residual, RMSNorm, gate|up, SwiGLU, down + residual] -> RMSNorm -> lm_head -> shifted CE
            (decode_fp32's prompt pass with the activations of every layer kept: residual stream, rotated q|k|v, attention output,
            gate|up)
  backward  dlogits (written by the CE kernel over the logits) -> lm_head dgrad -> 28 x [MLP, attention] dgrad through the frozen
            decoder (tasu_f32_gemm_nt on transposed fp32 weight copies; csrc/fp32_train.hip for RMSNorm / SwiGLU / attention / RoPE
            backward; attention probabilities are recomputed from the saved q|k|v) -> the audio rows' gradient -> projector weight
            gradients straight into the flat fp32 bucket ``proj.g`` (what TasuEngine's AdamW and the autograd boundary read).

A correctness mode: ~10x slower than the bf16 step at Qwen2.5-1.5B (271 against 28 ms per 16 utterances: the fp32 matrix rate is 1/16 of bf16's; the step runs at 0.61 of it); pinned on the
real reference's fp32 goldens (loss within 2e-5, projector gradients within 2e-4 relative L2: tests/test_gpu_model.py).  Decoder weights stay
frozen (dgrad only), like the bf16 step; LoRA and the non-default projectors train on the bf16 path only.
"""
import numpy as np
import torch

from .decode_fp32 import F32_MAX_CTX, _need_f32
from .model import HD, StepState, rup


def _transposed_weights(model):
    """fp32 W^T copies of the frozen decoder weights for the dgrad GEMMs (built once per model: +1x the fp32 weight bytes)."""
    llm = model.llm
    if llm.f32.get("t") is None:
        Vp = rup(model.geo.llm_vocab, 64)
        head = llm.f32["head"]
        head_t = torch.zeros(head.shape[1], Vp, dtype=torch.float32, device=head.device)
        head_t[:, : head.shape[0]].copy_(head.t())
        llm.f32["t"] = dict(layers=[{k: f[k].t().contiguous() for k in ("wqkv", "wo", "wgu", "wd")} for f in llm.f32["layers"]], head=head_t)
    return llm.f32["t"]


def forward_train_fp32(model, st: StepState):
    """Forward of the training step in fp32 with everything the backward needs kept in ``st.dev``; loss / accuracy in
    ``st.dev['loss_out']`` like the bf16 step; ``st.fp32 = True`` routes ``TasuModel.run_backward`` to ``backward_fp32``."""
    ops, geo, llm, pr = model.ops, model.geo, model.llm, model.proj
    _need_f32(model)
    if pr.kind != "linear-silu":
        raise NotImplementedError(f"the fp32 training step serves the shipped projector (linear-silu), not {pr.kind!r}")
    B, S, M = st.B, st.S, st.M
    if S > F32_MAX_CTX:
        raise ValueError(f"sequence length {S} exceeds the fp32 attention's limit {F32_MAX_CTX}")
    km = np.asarray(st.plan.key_mask)[:, :S].astype(bool)
    valid = km.sum(1).astype(np.int64)
    left = all(km[b, S - valid[b]:].all() for b in range(B))
    if not (left or all(km[b, :valid[b]].all() for b in range(B))):
        raise ValueError("the fp32 path expects every row's padding on one side")
    D, I, H, G, V, L = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads, geo.llm_vocab, geo.llm_layers
    LDQ, Vp = (H + 2 * G) * HD, rup(V, 64)
    scale = HD ** -0.5
    f32, i32 = torch.float32, torch.int32
    buf, d = model._buf, st.dev
    ws = buf("f32_gemm_ws", (16 * 128 * 4096,), f32)
    # ---- projector (EncoderProjectorLinearSiLU, projector.py:128-151) with its intermediates kept
    Fap, Rap, K, Kp, Hb, Do = st.Fap, st.Rap, pr.K, pr.Kp, pr.Hb, pr.Do
    if "post" not in d:
        post = buf("post", (Fap, Kp), f32)
        ops.posterior_build(d["post_ids"], d["post_alpha"], post, Fap, K)
        d["post"] = post
    xn_p = buf("f32t_xn_p", (Fap, Kp), f32)
    mean, rstd = buf("ln_mean", (Fap,), f32), buf("ln_rstd", (Fap,), f32)
    ops.layernorm_fwd(d["post"], pr.view(pr.p, "norm.weight"), pr.view(pr.p, "norm.bias"), xn_p, mean, rstd, Fap, K, geo.ln_eps)
    h_pre, h = buf("f32t_h_pre", (Rap, Hb), f32), buf("f32t_h", (Rap, Hb), f32)
    ops.f32_gemm(xn_p, pr.view(pr.p, pr.n_w1), h_pre, Rap, Hb, Kp, bias=pr.view(pr.p, pr.n_b1), ws=ws)
    ops.f32_silu(h_pre, h)
    y2 = buf("f32t_y2", (Rap, Do), f32)
    ops.f32_gemm(h, pr.view(pr.p, pr.n_w2), y2, Rap, Do, Hb, bias=pr.view(pr.p, pr.n_b2), ws=ws)
    # ---- decoder
    kstart = model._upload("f32_kstart_b", ((S - valid) if left else np.zeros(B, dtype=np.int64)).astype(np.int32))
    xs = buf("f32t_xs", (2 * L + 1, M, D), f32)                  # x_in[l] = xs[2l], x_mid[l] = xs[2l + 1], final = xs[2L]
    qkvs, aos = buf("f32t_qkv", (L, M, LDQ), f32), buf("f32t_ao", (L, M, H * HD), f32)
    gus = buf("f32t_gu", (L, M, 2 * I), f32)
    xn, act = buf("f32t_xn", (M, D), f32), buf("f32t_act", (M, I), f32)
    cos, sin = buf("f32_cos0", (M, HD // 2), f32), buf("f32_sin0", (M, HD // 2), f32)
    ops.f32_embed_merge(llm.embed, y2, d["kind"], d["idx"], xs[0], M, D)
    ops.rope_table(d["pos"], cos, sin, HD, geo.rope_theta)
    ops.f32_rmsnorm(xs[0], llm.layers[0]["ln1"], xn, M, D, geo.rms_eps)
    for l in range(L):
        f, w = llm.f32["layers"][l], llm.layers[l]
        next_norm = llm.layers[l + 1]["ln1"] if l + 1 < L else llm.norm
        ops.f32_gemm_qkv_rope(xn, f["wqkv"], f["bqkv"], qkvs[l], cos, sin, M, H, G, D, ws)
        ops.f32_attn_prefill(qkvs[l], kstart, aos[l], B, S, H, G, scale)
        ops.f32_gemm_resid_rmsnorm(aos[l], f["wo"], xs[2 * l + 1], w["ln2"], xn, M, D, H * HD, geo.rms_eps, ws, resid=xs[2 * l])
        ops.f32_gemm(xn, f["wgu"], gus[l], M, 2 * I, D, ws=ws)       # (gate|up is kept for the backward: not the fused finisher)
        ops.f32_swiglu(gus[l], act, M, I)
        ops.f32_gemm_resid_rmsnorm(act, f["wd"], xs[2 * l + 2], next_norm, xn, M, D, I, geo.rms_eps, ws, resid=xs[2 * l + 1])
    # ---- loss head: logits for every position (pad columns zeroed once: the lm_head dgrad contracts over Vp), CE + its gradient
    logits = buf("f32t_logits", (M, Vp), f32)
    ops.f32_gemm(xn, llm.f32["head"], logits, M, V, D, ws=ws)
    row_loss, row_hit = buf("row_loss", (M,), f32), buf("row_hit", (M,), i32)
    ops.f32_ce(logits, d["shift_labels"], M, V, row_loss, row_hit, dlogits=logits, inv_count=d["inv_count"])   # dlogits in place
    res = buf("loss_out", (4,), f32)
    ops.ce_reduce(row_loss, row_hit, d["shift_labels"], M, res)
    d.update(loss_out=res, f32t=dict(xs=xs, qkvs=qkvs, aos=aos, gus=gus, cos=cos, sin=sin, kstart=kstart, dlogits=logits, xn_p=xn_p, h_pre=h_pre,
                                     h=h, mean=mean, rstd=rstd))
    d.pop("logits", None)
    st.fp32 = True


def backward_fp32(model, st: StepState, on_ready=None):
    """dgrad through the frozen decoder and the projector's weight gradients into ``proj.g``, all fp32.  ``on_ready(lo, hi)``: the
    engine's gradient exchange hook, called once for the whole bucket at the end (this mode does not overlap the exchange)."""
    ops, geo, llm, pr = model.ops, model.geo, model.llm, model.proj
    B, S, M = st.B, st.S, st.M
    D, I, H, G, V, L = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads, geo.llm_vocab, geo.llm_layers
    LDQ, Vp = (H + 2 * G) * HD, rup(V, 64)
    scale = HD ** -0.5
    f32 = torch.float32
    buf, d = model._buf, st.dev
    a = d["f32t"]
    xs, qkvs, aos, gus = a["xs"], a["qkvs"], a["aos"], a["gus"]
    wt = _transposed_weights(model)
    ws = buf("f32_gemm_ws", (16 * 128 * 4096,), f32)
    dx, dn = buf("f32t_dx", (M, D), f32), buf("f32t_dn", (M, D), f32)
    dact, dgu = buf("f32t_dact", (M, I), f32), buf("f32t_dgu", (M, 2 * I), f32)
    dao, dqkv = buf("f32t_dao", (M, H * HD), f32), buf("f32t_dqkv", (M, LDQ), f32)
    lse, delta = buf("f32t_lse", (B * H * S,), f32), buf("f32t_delta", (B * H * S,), f32)
    # loss head
    ops.f32_gemm(a["dlogits"], wt["head"], dn, M, D, Vp, ws=ws)
    ops.f32_rmsnorm_bwd(dn, xs[2 * L], llm.norm, dx, M, D, geo.rms_eps, False)
    for l in range(L - 1, -1, -1):
        w, t = llm.layers[l], wt["layers"][l]
        # MLP block: x_out = x_mid + down(swiglu(gate|up(norm(x_mid))))
        ops.f32_gemm(dx, t["wd"], dact, M, I, D, ws=ws)
        ops.f32_swiglu_bwd(dact, gus[l], dgu, M, I)
        ops.f32_gemm(dgu, t["wgu"], dn, M, D, 2 * I, ws=ws)
        ops.f32_rmsnorm_bwd(dn, xs[2 * l + 1], w["ln2"], dx, M, D, geo.rms_eps, True)
        # attention block: x_mid = x_in + o(attention(rope(q|k|v(norm(x_in)))))
        ops.f32_gemm(dx, t["wo"], dao, M, H * HD, D, ws=ws)
        ops.f32_attn_bwd(qkvs[l], dao, a["kstart"], dqkv, lse, delta, B, S, H, G, scale)
        ops.f32_rope(dqkv, a["cos"], a["sin"], M, H, G, inverse=True)
        ops.f32_gemm(dqkv, t["wqkv"], dn, M, D, LDQ, ws=ws)
        ops.f32_rmsnorm_bwd(dn, xs[2 * l], w["ln1"], dx, M, D, geo.rms_eps, True)
    d["dx"] = dx
    if model.freeze_projector:
        return
    # ---- merge backward + projector backward (projector.py:149-151 reversed); weight gradients land in the flat bucket
    Fap, Rap, K, Kp, Hb, Do = st.Fap, st.Rap, pr.K, pr.Kp, pr.Hb, pr.Do
    rows = d["audio_rows_pad"] if "audio_rows_pad" in d else model._pad_rows(st)
    dy2 = buf("f32t_dy2", (Rap, Do), f32)
    ops.f32_gather_rows(dx, rows, dy2, Rap, Do)
    ops.f32_colsum(dy2, pr.view(pr.g, pr.n_b2), Rap, Do)
    dy2_t, h_t = buf("f32t_dy2_t", (Do, Rap), f32), buf("f32t_h_t", (Hb, Rap), f32)
    ops.f32_transpose(dy2, dy2_t, Rap, Do, Rap)
    ops.f32_transpose(a["h"], h_t, Rap, Hb, Rap)
    ops.f32_gemm(dy2_t, h_t, pr.view(pr.g, pr.n_w2), Do, Hb, Rap, ws=ws)                    # dW2 = dy2^T h
    w2_t = buf("f32t_w2_t", (Hb, Do), f32)
    ops.f32_transpose(pr.view(pr.p, pr.n_w2), w2_t, Do, Hb, Do)
    dh = buf("f32t_dh", (Rap, Hb), f32)
    ops.f32_gemm(dy2, w2_t, dh, Rap, Hb, Do, ws=ws)
    ops.f32_silu(a["h_pre"], dh, dy=dh)                                                      # dh_pre, in place
    ops.f32_colsum(dh, pr.view(pr.g, pr.n_b1), Rap, Hb)
    dh_t, xn_t = buf("f32t_dh_t", (Hb, Rap), f32), buf("f32t_xn_t", (Kp, Rap), f32)
    ops.f32_transpose(dh, dh_t, Rap, Hb, Rap)
    ops.f32_transpose(a["xn_p"], xn_t, Rap, Kp, Rap)
    ops.f32_gemm(dh_t, xn_t, pr.view(pr.g, pr.n_w1), Hb, Kp, Rap, ws=ws)                     # dW1 = dh_pre^T xn
    w1_t = buf("f32t_w1_t", (Kp, Hb), f32)
    ops.f32_transpose(pr.view(pr.p, pr.n_w1), w1_t, Hb, Kp, Hb)
    dxn = buf("f32t_dxn", (Rap, Kp), f32)
    ops.f32_gemm(dh, w1_t, dxn, Rap, Kp, Hb, ws=ws)
    ops.f32_layernorm_bwd_params(dxn, d["post"], a["mean"], a["rstd"], pr.view(pr.g, "norm.weight"), pr.view(pr.g, "norm.bias"), Rap, K)
    if on_ready is not None:
        on_ready(0, pr.numel)
