"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch-CPU fp32 tensors, no GPU, no HIP) of the TASU
hot path: SenseVoiceSmall encoder -> CTC posterior -> PSD / text pseudo-posterior -> LinearSiLU projector
-> merge -> Qwen2 decoder -> shifted CE + token accuracy, plus AdamW / WarmupCosineLR.

It is the *checker* for the HIP path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
The product (ps_slm_amd/) never imports this file.

PINNING: every function here is checked in tests/test_oracle_golden.py against fixtures produced by
oracle/make_golden.py from the REAL reference modules (/root/reference/Multitask/model/{ps-slm,SenseVoice,
projector}.py + transformers.Qwen2ForCausalLM) on seeded random-init weights.  The optimizer/schedule of
the reference lives in DeepSpeed (absent here and unpinned in the reference's Dockerfile): AdamW is pinned
against torch.optim.AdamW; WarmupCosineLR is restated from DeepSpeed's published formula -> "parity
unpinned" for that one function.

Weights are a flat dict with the reference's state-dict names (``encoder.*``, ``encoder_projector.*``,
``llm.*``).  ``mode``: "fp32" = the shipped default numerics; "bf16" = emulation of
``torch.autocast(dtype=bfloat16)`` (Multitask/utils/deepspeed_utils.py:160): linear/matmul inputs and
outputs rounded to bf16, fp32 accumulation, norms / softmax / residual stream / loss in fp32.
"""
import math

import torch
import torch.nn.functional as F

IGNORE = -100


# ----------------------------------------------------------------------------- rounding helpers
class _RoundBF16(torch.autograd.Function):
    """bf16 round trip whose gradient is rounded to bf16 too (what autocast does to a bf16 tensor)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def rbf(x, mode):
    return _RoundBF16.apply(x) if mode == "bf16" else x


def linear(x, w, b, mode):
    """autocast nn.Linear: bf16 operands, fp32 accumulate, bf16 result."""
    y = rbf(x, mode) @ rbf(w, mode).t()
    if b is not None:
        y = y + rbf(b, mode)
    return rbf(y, mode)


# ----------------------------------------------------------------------------- SenseVoice encoder
def sinusoidal_pe(T, depth, dtype=torch.float32):
    """Multitask/model/SenseVoice.py:26-50: positions 1..T, [sin | cos], increment log(1e4)/(depth/2-1)."""
    pos = torch.arange(1, T + 1, dtype=dtype)
    inc = math.log(10000.0) / (depth / 2 - 1)
    inv = torch.exp(torch.arange(depth // 2, dtype=dtype) * (-inc))
    st = pos[:, None] * inv[None, :]
    return torch.cat([torch.sin(st), torch.cos(st)], dim=1)  # [T, depth]


def layer_norm(x, w, b, eps=1e-5):
    """fp32 LayerNorm (SenseVoice.py:270-282 forces fp32; projector LN is fp32 under autocast too)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def sanm_attention(x, maskf, W, pre, heads, ksize, mode):
    """MultiHeadedAttentionSANM.forward, SenseVoice.py:209-228 (+ :124-207).  maskf: [B,T] float 0/1."""
    B, T, _ = x.shape
    qkv = linear(x, W[pre + "linear_q_k_v.weight"], W[pre + "linear_q_k_v.bias"], mode)
    D = qkv.shape[-1] // 3
    dk = D // heads
    q, k, v = qkv.split(D, dim=-1)
    # FSMN memory block on the un-split v (:124-140): mask, pad (k-1)/2 each side, depthwise conv, + v, mask
    vm = v * maskf[:, :, None]
    left = (ksize - 1) // 2
    xp = F.pad(vm.transpose(1, 2), (left, ksize - 1 - left))
    fs = F.conv1d(xp, W[pre + "fsmn_block.weight"], None, groups=D).transpose(1, 2)
    fsmn = (fs + vm) * maskf[:, :, None]
    # attention (:171-207, :225-227)
    qh = q.reshape(B, T, heads, dk).transpose(1, 2) * dk ** (-0.5)
    kh = k.reshape(B, T, heads, dk).transpose(1, 2)
    vh = v.reshape(B, T, heads, dk).transpose(1, 2)
    scores = rbf(rbf(qh, mode) @ rbf(kh, mode).transpose(-1, -2), mode)
    dead = (maskf == 0)[:, None, None, :]
    scores = scores.masked_fill(dead, float("-inf"))
    attn = torch.softmax(scores, dim=-1).masked_fill(dead, 0.0)
    ctx = rbf(rbf(attn, mode) @ rbf(vh, mode), mode).transpose(1, 2).reshape(B, T, D)
    out = linear(ctx, W[pre + "linear_out.weight"], W[pre + "linear_out.bias"], mode)
    return out + fsmn


def sanm_layer(x, maskf, W, pre, heads, ksize, mode):
    """EncoderLayerSANM.forward (pre-norm, eval), SenseVoice.py:324-400."""
    in_size = W[pre + "norm1.weight"].shape[0]
    size = W[pre + "norm2.weight"].shape[0]
    h = layer_norm(x, W[pre + "norm1.weight"], W[pre + "norm1.bias"])
    a = sanm_attention(h, maskf, W, pre + "self_attn.", heads, ksize, mode)
    x = x + a if in_size == size else a
    h = layer_norm(x, W[pre + "norm2.weight"], W[pre + "norm2.bias"])
    f = linear(h, W[pre + "feed_forward.w_1.weight"], W[pre + "feed_forward.w_1.bias"], mode)
    f = linear(torch.relu(f), W[pre + "feed_forward.w_2.weight"], W[pre + "feed_forward.w_2.bias"], mode)
    return x + f


def sensevoice_encoder(W, feats, lens, heads, ksize, mode="fp32", pre="encoder.encoder."):
    """SenseVoiceEncoderSmall.forward, SenseVoice.py:548-579.  feats [B,T,F] (NOT modified), lens [B]."""
    B, T, Fd = feats.shape
    maskf = (torch.arange(T)[None, :] < lens[:, None]).float()
    out_size = W[pre + "after_norm.weight"].shape[0]
    x = feats * out_size ** 0.5 + sinusoidal_pe(T, Fd)[None]
    n0 = len({k.split(".")[3] for k in W if k.startswith(pre + "encoders0.")})
    n1 = len({k.split(".")[3] for k in W if k.startswith(pre + "encoders.")})
    n2 = len({k.split(".")[3] for k in W if k.startswith(pre + "tp_encoders.")})
    for i in range(n0):
        x = sanm_layer(x, maskf, W, f"{pre}encoders0.{i}.", heads, ksize, mode)
    for i in range(n1):
        x = sanm_layer(x, maskf, W, f"{pre}encoders.{i}.", heads, ksize, mode)
    x = layer_norm(x, W[pre + "after_norm.weight"], W[pre + "after_norm.bias"])
    for i in range(n2):
        x = sanm_layer(x, maskf, W, f"{pre}tp_encoders.{i}.", heads, ksize, mode)
    x = layer_norm(x, W[pre + "tp_norm.weight"], W[pre + "tp_norm.bias"])
    return x, maskf.sum(1).to(torch.int32)


QUERY_ROWS = (0, 1, 2, 2)  # language(0), event/emo(1,2), textnorm(2): ps-slm.py:430-442


def audio_front(W, feats, feat_lens, heads, ksize, mode="fp32"):
    """ps-slm.py:430-454: prepend the 4 query embeddings, run encoder + CTC head + softmax, drop 4 frames."""
    B = feats.shape[0]
    q = W["encoder.embed.weight"][list(QUERY_ROWS)][None].expand(B, -1, -1)
    speech = torch.cat([q, feats.to(q.dtype)], dim=1)
    enc, olens = sensevoice_encoder(W, speech, feat_lens + 4, heads, ksize, mode)
    logits = linear(enc, W["encoder.ctc.ctc_lo.weight"], W["encoder.ctc.ctc_lo.bias"], mode)
    post = torch.softmax(logits, dim=-1)
    return post[:, 4:], enc[:, 4:], torch.clamp(olens - 4, min=0)


def psd(feats, lens, posterior, blank_id=0, blank_threshold=0.90):
    """ps-slm.py:237-317.  Per utterance: ids = argmax; runs of equal NON-blank ids collapse to their mean
    row (blank frames stay single); item blank-prob = mean P[t, blank]; keep items with prob < threshold;
    zero-pad to the batch max.  Returns ([B, T', D], [B] int64)."""
    B, T, D = feats.shape
    probs = posterior.exp() if bool(posterior.max() <= 0) else posterior
    kept = []
    for b in range(B):
        L = int(lens[b])
        if L == 0:
            kept.append(feats.new_zeros(0, D))
            continue
        ids = probs[b, :L].argmax(-1)
        # segment boundaries: a new segment starts at t if id changed, or the id is blank
        start = torch.ones(L, dtype=torch.bool)
        start[1:] = (ids[1:] != ids[:-1]) | (ids[1:] == blank_id)
        seg = torch.cumsum(start.long(), 0) - 1
        nseg = int(seg[-1]) + 1
        cnt = torch.zeros(nseg).index_add_(0, seg, torch.ones(L))
        rows = []
        bp = []
        # mean over the run, computed the way the reference does (tensor.mean over the slice)
        bounds = torch.nonzero(start).flatten().tolist() + [L]
        for s in range(nseg):
            a, e = bounds[s], bounds[s + 1]
            if e - a == 1:
                rows.append(feats[b, a])
                bp.append(probs[b, a, blank_id])
            else:
                rows.append(feats[b, a:e].mean(dim=0))
                bp.append(probs[b, a:e, blank_id].mean())
        del cnt
        rows = torch.stack(rows)
        bp = torch.stack(bp)
        kept.append(rows[bp < blank_threshold])
    new_lens = torch.tensor([k.shape[0] for k in kept], dtype=torch.long)
    mx = int(new_lens.max()) if B else 0
    if mx == 0:
        return feats.new_zeros(B, 0, D), torch.zeros(B, dtype=torch.long)
    out = feats.new_zeros(B, mx, D)
    for b, k in enumerate(kept):
        out[b, : k.shape[0]] = k
    return out, new_lens


def pseudo_posterior(ids_list, V, alphas=None, keeps=None):
    """ps-slm.py:337-358 (clean: alphas=None) and :360-409 (noise; insert_prob = 0).  The random draws
    (alpha per utterance, keep mask per token) are explicit inputs."""
    rows = []
    for u, ids in enumerate(ids_list):
        ids = torch.as_tensor(ids, dtype=torch.long)
        oh = F.one_hot(ids, V).float() if len(ids) else torch.zeros(0, V)
        if alphas is not None:
            a = float(alphas[u])
            oh = (1 - a) * oh + a / V
            oh = oh[torch.as_tensor(keeps[u], dtype=torch.bool)]
        rows.append(oh)
    lens = torch.tensor([r.shape[0] for r in rows], dtype=torch.long)
    out = torch.zeros(len(rows), int(lens.max()), V)
    for b, r in enumerate(rows):
        out[b, : r.shape[0]] = r
    return out, lens


# ----------------------------------------------------------------------------- projector
def projector(W, x, mode="fp32", pre="encoder_projector."):
    """EncoderProjectorLinearSiLU.forward, Multitask/model/projector.py:139-151; with ``linear1.*`` keys instead:
    EncoderProjectorConcat.forward, :38-49 (k = in_features / feature width frames concatenated, ReLU, no norm); with
    ``conv1d.*`` keys as well: EncoderProjectorCov1d.forward, :64-73 (Conv1d kernel = stride = k over time, ReLU, Linear, ReLU,
    Linear; autocast runs the convolution in bf16 like a Linear)."""
    if pre + "W_q.weight" in W:
        # EncoderProjectorCTCCA.forward, projector.py:111-126 (called with the detached input-embedding table, ps-slm.py:475-479);
        # under autocast: Linear and both einsums in bf16, the scale on the bf16 scores, softmax in fp32
        E = W["llm.model.embed_tokens.weight"].detach()
        B, T, _ = x.shape
        Q = linear(x, W[pre + "W_q.weight"], None, mode)
        h = 8
        d = Q.shape[-1] // h
        q = Q.view(B, T, h, d)
        k = rbf(E, mode).view(-1, h, d)
        scores = rbf(rbf(torch.einsum("bthd,vhd->bthv", rbf(q, mode), k), mode) / d ** 0.5, mode)
        attn = scores.softmax(dim=-1)
        z = rbf(torch.einsum("bthv,vhd->bthd", rbf(attn, mode), k), mode)
        return z.reshape(B, T, -1)
    if pre + "conv1d.weight" in W:
        w = W[pre + "conv1d.weight"]                         # [out, in, k]
        k = w.shape[2]
        c = F.conv1d(rbf(x, mode).transpose(1, 2), rbf(w, mode), rbf(W[pre + "conv1d.bias"], mode), stride=k).transpose(1, 2)
        h = rbf(F.relu(rbf(c, mode)), mode)
        h = rbf(F.relu(linear(h, W[pre + "linear1.weight"], W[pre + "linear1.bias"], mode)), mode)
        return linear(h, W[pre + "linear2.weight"], W[pre + "linear2.bias"], mode)
    if pre + "linear1.weight" in W:
        k = W[pre + "linear1.weight"].shape[1] // x.shape[-1]
        B, T, Dm = x.shape
        x = x[:, : (T // k) * k].reshape(B, T // k, Dm * k)
        h = rbf(F.relu(linear(x, W[pre + "linear1.weight"], W[pre + "linear1.bias"], mode)), mode)
        return linear(h, W[pre + "linear2.weight"], W[pre + "linear2.bias"], mode)
    h = layer_norm(x, W[pre + "norm.weight"], W[pre + "norm.bias"], 1e-5)
    h = linear(h, W[pre + "ffn.0.weight"], W[pre + "ffn.0.bias"], mode)
    h = rbf(F.silu(h), mode)
    return linear(h, W[pre + "ffn.2.weight"], W[pre + "ffn.2.bias"], mode)


# ----------------------------------------------------------------------------- merge
def merge_plan(input_ids, attention_mask, num_audio, speech_id):
    """Index plan of _merge_input_ids_with_audio_features, ps-slm.py:679-873.

    Returns dict with S (merged length), left_padding, text_dst [B,L] (destination column of every input
    column, -1 if not copied), audio_dst [B] (first destination column of the audio span), final mask
    [B,S] bool, position_ids [B,S] int64."""
    B, L = input_ids.shape
    am = attention_mask.bool()
    lp = bool((~am[:, 0]).any())
    rp = bool((~am[:, -1]).any())
    left = True
    if B > 1:
        if lp and rp:
            raise ValueError(f"both side of attention_mask has zero, invalid. {attention_mask}")
        left = not (rp and not lp)
    is_sp = input_ids == speech_id
    width = torch.ones_like(input_ids)
    width[is_sp] = num_audio.long()  # one <speech> per row, row order (ps-slm.py:805-807)
    new_pos = torch.cumsum(width, -1) - 1
    tot = width.sum(-1)
    S = int(tot.max())
    if left:
        new_pos = new_pos + (S - 1 - new_pos[:, -1])[:, None]
    is_text = (~is_sp) & am
    text_dst = torch.where(is_text, new_pos, torch.full_like(new_pos, -1))
    npad = (~am).long().sum(-1)
    real = tot - npad  # number of real (text + audio) slots per row
    col = torch.arange(S)[None, :]
    live = (S - col) <= real[:, None] if left else col < real[:, None]
    text_slot = torch.zeros(B, S, dtype=torch.bool)
    bi, li = torch.nonzero(is_text, as_tuple=True)
    text_slot[bi, text_dst[bi, li]] = True
    audio_slot = live & ~text_slot
    if int(audio_slot.sum()) != int(num_audio.sum()):
        raise ValueError("The input provided to the model are wrong: audio slots != audio tokens")
    mask = text_slot | audio_slot
    pos = (mask.long().cumsum(-1) - 1).masked_fill(~mask, 1)
    return dict(S=S, left_padding=left, text_dst=text_dst, audio_slot=audio_slot, mask=mask, position_ids=pos)


def merge(audio_feats, num_audio, tok_embeds, input_ids, attention_mask, labels, speech_id):
    plan = merge_plan(input_ids, attention_mask, num_audio, speech_id)
    B, L = input_ids.shape
    S = plan["S"]
    D = tok_embeds.shape[-1]
    emb = torch.zeros(B, S, D, dtype=tok_embeds.dtype)
    bi, li = torch.nonzero(plan["text_dst"] >= 0, as_tuple=True)
    di = plan["text_dst"][bi, li]
    emb = emb.index_put((bi, di), tok_embeds[bi, li])
    amask = torch.arange(audio_feats.shape[1])[None, :] < num_audio[:, None]
    emb = emb.masked_scatter(plan["audio_slot"][:, :, None], audio_feats[amask].to(emb.dtype))
    lab = None
    if labels is not None:
        lab = torch.full((B, S), IGNORE, dtype=torch.long)
        lab[bi, di] = labels[bi, li]
    return emb, plan["mask"], lab, plan["position_ids"]


# ----------------------------------------------------------------------------- Qwen2 decoder
def rms_norm(x, w, eps=1e-6):
    """Qwen2RMSNorm: fp32, x * rsqrt(mean x^2 + eps) * w."""
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def rope_tables(position_ids, head_dim, theta):
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = position_ids[:, :, None].float() * inv[None, None, :]
    emb = torch.cat([fr, fr], -1)
    return emb.cos(), emb.sin()  # [B,S,hd]


def apply_rope(x, cos, sin):
    """x [B,H,S,hd]; rotate-half convention."""
    h = x.shape[-1] // 2
    rot = torch.cat([-x[..., h:], x[..., :h]], -1)
    return x * cos[:, None] + rot * sin[:, None]


def qwen2_geometry(W, pre="llm."):
    nl = len({k.split(".")[3] for k in W if k.startswith(pre + "model.layers.")})
    return nl


def qwen2_hidden(W, emb, mask, position_ids, n_heads, n_kv, theta=1e6, mode="fp32", pre="llm.",
                 return_kv=False):
    """Qwen2Model.forward restated (transformers modeling_qwen2.py): 28x[RMSNorm, qkv(+bias), RoPE,
    causal GQA attention with key-padding, o_proj, +res, RMSNorm, SwiGLU, +res], final RMSNorm."""
    B, S, Dm = emb.shape
    hd = W[pre + "model.layers.0.self_attn.q_proj.weight"].shape[0] // n_heads
    cos, sin = rope_tables(position_ids, hd, theta)
    allow = torch.tril(torch.ones(S, S, dtype=torch.bool))[None, None] & mask.bool()[:, None, None, :]
    x = emb
    kvs = []
    for l in range(qwen2_geometry(W, pre)):
        p = f"{pre}model.layers.{l}."
        h = rms_norm(x, W[p + "input_layernorm.weight"])
        q = linear(h, W[p + "self_attn.q_proj.weight"], W[p + "self_attn.q_proj.bias"], mode)
        k = linear(h, W[p + "self_attn.k_proj.weight"], W[p + "self_attn.k_proj.bias"], mode)
        v = linear(h, W[p + "self_attn.v_proj.weight"], W[p + "self_attn.v_proj.bias"], mode)
        q = rbf(apply_rope(q.view(B, S, n_heads, hd).transpose(1, 2), cos, sin), mode)
        k = rbf(apply_rope(k.view(B, S, n_kv, hd).transpose(1, 2), cos, sin), mode)
        v = v.view(B, S, n_kv, hd).transpose(1, 2)
        kvs.append((k, v))
        rep = n_heads // n_kv
        kk = k.repeat_interleave(rep, dim=1)
        vv = v.repeat_interleave(rep, dim=1)
        sc = (q @ kk.transpose(-1, -2)) * hd ** (-0.5)
        sc = sc.masked_fill(~allow, float("-inf"))
        pr = torch.softmax(sc, dim=-1)
        pr = torch.nan_to_num(pr, nan=0.0)  # fully-masked (left-pad) query rows: defined as 0
        a = rbf(rbf(pr, mode) @ vv, mode).transpose(1, 2).reshape(B, S, n_heads * hd)
        x = x + linear(a, W[p + "self_attn.o_proj.weight"], None, mode)
        h = rms_norm(x, W[p + "post_attention_layernorm.weight"])
        g = linear(h, W[p + "mlp.gate_proj.weight"], None, mode)
        u = linear(h, W[p + "mlp.up_proj.weight"], None, mode)
        act = rbf(rbf(F.silu(g), mode) * u, mode)
        x = x + linear(act, W[p + "mlp.down_proj.weight"], None, mode)
    x = rms_norm(x, W[pre + "model.norm.weight"])
    return (x, kvs) if return_kv else x


def qwen2_hidden_step(W, x, kvs, key_mask, position_ids, n_heads, n_kv, theta=1e6, mode="fp32", pre="llm."):
    """One decode position with a KV cache -- what HF ``generate`` runs per step (``use_cache=True`` is transformers' default; the
    reference's call, Multitask/model/ps-slm.py:660-675, does not turn it off): x [R, 1, D] is the new token's embedding,
    ``kvs[l] = (k, v)`` the rotated keys / values of the earlier positions [R, n_kv, T, hd], ``key_mask`` [R, T + 1] (prompt padding
    + the new position), ``position_ids`` [R, 1].  Same arithmetic per row as ``qwen2_hidden``'s last position.  Returns
    (hidden [R, 1, D], the extended cache)."""
    R = x.shape[0]
    hd = W[pre + "model.layers.0.self_attn.q_proj.weight"].shape[0] // n_heads
    cos, sin = rope_tables(position_ids, hd, theta)
    allow = key_mask.bool()[:, None, None, :]
    rep = n_heads // n_kv
    out = []
    for l in range(qwen2_geometry(W, pre)):
        p = f"{pre}model.layers.{l}."
        h = rms_norm(x, W[p + "input_layernorm.weight"])
        q = linear(h, W[p + "self_attn.q_proj.weight"], W[p + "self_attn.q_proj.bias"], mode)
        k = linear(h, W[p + "self_attn.k_proj.weight"], W[p + "self_attn.k_proj.bias"], mode)
        v = linear(h, W[p + "self_attn.v_proj.weight"], W[p + "self_attn.v_proj.bias"], mode)
        q = rbf(apply_rope(q.view(R, 1, n_heads, hd).transpose(1, 2), cos, sin), mode)
        k = torch.cat([kvs[l][0], rbf(apply_rope(k.view(R, 1, n_kv, hd).transpose(1, 2), cos, sin), mode)], 2)
        v = torch.cat([kvs[l][1], v.view(R, 1, n_kv, hd).transpose(1, 2)], 2)
        out.append((k, v))
        sc = (q @ k.repeat_interleave(rep, dim=1).transpose(-1, -2)) * hd ** (-0.5)
        pr = torch.softmax(sc.masked_fill(~allow, float("-inf")), dim=-1)
        a = rbf(rbf(pr, mode) @ v.repeat_interleave(rep, dim=1), mode).transpose(1, 2).reshape(R, 1, n_heads * hd)
        x = x + linear(a, W[p + "self_attn.o_proj.weight"], None, mode)
        h = rms_norm(x, W[p + "post_attention_layernorm.weight"])
        g = linear(h, W[p + "mlp.gate_proj.weight"], None, mode)
        u = linear(h, W[p + "mlp.up_proj.weight"], None, mode)
        x = x + linear(rbf(rbf(F.silu(g), mode) * u, mode), W[p + "mlp.down_proj.weight"], None, mode)
    return rms_norm(x, W[pre + "model.norm.weight"]), out


def lm_head_weight(W, pre="llm."):
    return W[pre + "lm_head.weight"] if (pre + "lm_head.weight") in W else W[pre + "model.embed_tokens.weight"]


def causal_lm_loss(logits, labels):
    """transformers loss_utils.ForCausalLMLoss: fp32 logits, labels shifted left, mean CE over != -100."""
    B, S, V = logits.shape
    shift = torch.cat([labels[:, 1:], torch.full((B, 1), IGNORE, dtype=labels.dtype)], 1)
    return F.cross_entropy(logits.float().reshape(-1, V), shift.reshape(-1), ignore_index=IGNORE)


def token_accuracy(logits, labels):
    """ps-slm.py:533-535 + Multitask/utils/metric.py:3-20."""
    pred = logits.argmax(-1)[:, :-1]
    tgt = labels[:, 1:]
    m = tgt != IGNORE
    return ((pred == tgt) & m).sum().float() / m.sum().float()


# ----------------------------------------------------------------------------- whole forward
def forward_text(W, batch, geo, mode="fp32"):
    """slam_model_asr.forward, text-only branch (gt_emb): ps-slm.py:459-468,482,525-535.
    batch: input_ids, attention_mask, labels, post_ids (list of id lists), optional alphas/keeps."""
    post, plen = pseudo_posterior(batch["post_ids"], geo["ctc_vocab"], batch.get("alphas"), batch.get("keeps"))
    return forward_from_posterior(W, batch, post, plen, geo, mode)


def forward_audio(W, batch, geo, mode="fp32", raw=False):
    """slam_model_asr.forward, audio branch (gt_emb=false, do_psd): ps-slm.py:430-454,471; ``raw``: the ctc_posterior=false
    branch (ps-slm.py:515-523): PSD's decisions from the posterior, its rows from the encoder's output states."""
    post, enc, lens = audio_front(W, batch["input_features"], batch["input_feature_length"], geo["enc_heads"],
                                  geo["enc_kernel"], mode)
    x, plen = psd(enc if raw else post, lens, post, blank_id=0)
    return forward_from_posterior(W, batch, x, plen, geo, mode)


def forward_from_posterior(W, batch, post, plen, geo, mode="fp32"):
    proj = projector(W, post, mode)
    if "encoder_projector.conv1d.weight" in W:             # Conv1d kernel = stride = k: len // k rows (ps-slm.py:482)
        plen = plen // W["encoder_projector.conv1d.weight"].shape[2]
    elif "encoder_projector.linear1.weight" in W:          # k frames per projector row: len // k rows (ps-slm.py:482)
        plen = plen // (W["encoder_projector.linear1.weight"].shape[1] // post.shape[-1])
    tok = W["llm.model.embed_tokens.weight"][batch["input_ids"]]
    emb, mask, lab, pos = merge(proj, plen, tok, batch["input_ids"], batch["attention_mask"],
                                batch.get("labels"), geo["speech_id"])
    hid = qwen2_hidden(W, emb, mask, pos, geo["llm_heads"], geo["llm_kv_heads"], geo.get("rope_theta", 1e6), mode)
    logits = linear(hid, lm_head_weight(W), None, mode)
    out = dict(logits=logits, mask=mask, labels=lab, position_ids=pos, embeds=emb, proj=proj)
    if lab is not None:
        out["loss"] = causal_lm_loss(logits, lab)
        out["acc"] = token_accuracy(logits, lab)
    return out


PROJ_KEYS = tuple("encoder_projector." + k for k in
                  ("norm.weight", "norm.bias", "ffn.0.weight", "ffn.0.bias", "ffn.2.weight", "ffn.2.bias"))


def loss_and_projector_grads(W, batch, geo, mode="fp32", audio=False, raw=False):
    keys = tuple(k for k in W if k.startswith("encoder_projector."))      # either projector kind
    Wg = dict(W)
    for k in keys:
        Wg[k] = W[k].detach().clone().requires_grad_(True)
    out = forward_audio(Wg, batch, geo, mode, raw=raw) if audio else forward_text(Wg, batch, geo, mode)
    grads = torch.autograd.grad(out["loss"], [Wg[k] for k in keys])
    return out, dict(zip(keys, grads))


# ----------------------------------------------------------------------------- optimizer / schedule
def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.0):
    """Decoupled-weight-decay Adam with bias correction (DeepSpeed FusedAdam adam_w_mode=True, the
    optimizer of Multitask/conf/ds_config.json:4-11; == torch.optim.AdamW).  step is 1-based.  In place."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    p.mul_(1 - lr * weight_decay)
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def warmup_cosine_lr(it, base_lr, warmup_num_steps=200, total_num_steps=15000, warmup_min_ratio=0.0,
                     cos_min_ratio=1e-4, warmup_type="log"):
    """DeepSpeed WarmupCosineLR (ds_config.json:22-27).  DeepSpeed is not in the reference tree and not
    installed here -> restated from its published lr_schedules.py: PARITY UNPINNED.
    ``it`` = the scheduler's ``last_batch_iteration``; it < 0 (before the first scheduler.step()) -> 0."""
    warmup_num_steps = max(2, warmup_num_steps)
    if it < 0:
        return 0.0
    if it < warmup_num_steps:
        r = math.log(it + 1) / math.log(warmup_num_steps) if warmup_type == "log" else it / warmup_num_steps
        ratio = warmup_min_ratio + (1.0 - warmup_min_ratio) * r
    else:
        real_last = it - warmup_num_steps + 1
        real_total = total_num_steps - warmup_num_steps
        ratio = max(0.0, cos_min_ratio + (1 - cos_min_ratio) * 0.5 * (1 + math.cos(math.pi * real_last / real_total)))
    return base_lr * ratio


def lr_for_optimizer_step(k, base_lr, **kw):
    """LR in effect for the k-th (1-based) optimizer step: the DeepSpeed engine calls optimizer.step()
    and then lr_scheduler.step(), and the scheduler starts at last_batch_iteration = -1, so step k sees
    the ratio of iteration k-2 (steps 1 and 2 run at lr 0 with log warmup).  PARITY UNPINNED."""
    return warmup_cosine_lr(k - 2, base_lr, **kw)


# ----------------------------------------------------------------------------- decode: beam search
def bf16_ulp_jitter(seed, prob=0.15):
    """Returns a ``logit_jitter`` for beam_search_generate: each logit moves by +-1 bf16 ulp with probability ``prob`` (two
    bf16 evaluations of one network differ like this wherever an fp32 sum sits next to a rounding boundary, at a few
    per cent of the logits)."""
    g = torch.Generator().manual_seed(seed)

    def jitter(logits):
        ulp = torch.exp2(torch.floor(torch.log2(logits.abs().clamp_min(1e-30))) - 7)
        flip = (torch.rand(logits.shape, generator=g) < prob).float() * (torch.randint(0, 2, logits.shape, generator=g) * 2 - 1)
        return logits + flip * ulp
    return jitter


def beam_search_generate(W, emb, mask, geo, num_beams=4, max_new_tokens=200, min_length=1, length_penalty=1.0,
                         eos_token_id=None, pad_token_id=None, mode="fp32", logit_jitter=None, logits_trace=None,
                         logits_replay=None, kv_cache=False, step_times=None):
    """slam_model_asr.generate's decode loop (Multitask/model/ps-slm.py:660-675): HF ``generate(inputs_embeds=...,
    num_beams=4, do_sample=False, early_stopping=False)`` restated (transformers generation/utils.py ``_beam_search``,
    un-vendored dependency): every step keeps the 2*num_beams best continuations, the first num_beams non-finished
    ones keep running, finished ones (EOS or max length) among the top num_beams compete for the num_beams result
    slots with score / (generated_length ** length_penalty); the loop ends when no running beam can beat the worst
    kept result (heuristic on the current length) or every continuation hit a stopping criterion.  By default the whole
    sequence is re-run every step (the parity fixtures: tiny sizes, one code path for the network); ``kv_cache=True`` runs the
    prompt once and then ``qwen2_hidden_step`` per position on a cache that follows the beams (what HF generate does: the CPU
    baseline of bench.py times this form; ``step_times``, a list, receives the wall-clock time after every position's network
    pass).  Returns new tokens only, [B, n_new].

    ``logit_jitter`` (optional callable logits -> logits) perturbs every step's logits; the fixture generator
    oracle/make_golden_generate_margin.py uses it to keep only decode cases whose tokens survive random one-ulp flips of the
    bf16 logits (``bf16_ulp_jitter``), i.e. cases without near-ties.  ``logits_trace`` (a list) receives every step's
    (logits, running prefixes); ``logits_replay`` (such a list) replaces the network: a step whose running prefixes equal the
    recorded ones reuses the recorded logits (the network is a function of the prefixes), any other step ends the call with
    ``None`` -- a jittered run over a recorded trajectory therefore costs only the bookkeeping, and a run that leaves the
    trajectory has already shown the case to be unstable."""
    B, S, D = emb.shape
    nb, V = num_beams, lm_head_weight(W).shape[0]
    # HF counts ``min_length`` INCLUDING the prompt, and under ``inputs_embeds`` subtracts the embedded prompt's length from it
    # (transformers generation/utils.py GenerationMixin._prepare_generated_length: ``min_length = max(min_length -
    # inputs_tensor.shape[1], 0)``; the reference passes inputs_embeds + min_length, ps-slm.py:660-668): EOS is banned only for
    # the first max(min_length - S, 0) generated positions -- with the reference's default min_length = 1 never.  (Pinned by
    # tests/golden/mid_generate_fp32.npz; rounds 1-5 banned the first min_length positions.)
    min_length = max(int(min_length) - S, 0)
    eos = geo["eos_id"] if eos_token_id is None else eos_token_id
    pad = eos if pad_token_id is None else pad_token_id
    K = 2 * nb
    NEG = -1.0e9
    run_seq = torch.full((B, nb, max_new_tokens), pad, dtype=torch.long)
    fin_seq = run_seq.clone()
    run_scores = torch.zeros(B, nb)
    run_scores[:, 1:] = NEG
    fin_scores = torch.full((B, nb), NEG)
    fin_len = torch.zeros(B, nb, dtype=torch.long)
    is_fin = torch.zeros(B, nb, dtype=torch.bool)
    unsat = torch.ones(B, 1, dtype=torch.bool)
    top_mask = torch.cat([torch.ones(nb, dtype=torch.bool), torch.zeros(K - nb, dtype=torch.bool)])
    table = W["llm.model.embed_tokens.weight"]
    emb_b = emb.repeat_interleave(nb, 0)
    mask_b = mask.repeat_interleave(nb, 0)
    cur = 0
    while True:
        toks = run_seq.view(B * nb, -1)[:, :cur]
        if logits_replay is not None:
            if cur >= len(logits_replay) or not torch.equal(logits_replay[cur][1], toks):
                return None
            logits = logits_replay[cur][0].clone()
        elif kv_cache:
            m = torch.cat([mask_b.bool(), torch.ones(B * nb, cur, dtype=torch.bool)], 1)
            if cur == 0:
                pos = (m.long().cumsum(-1) - 1).masked_fill(~m, 1)
                hid, cache = qwen2_hidden(W, emb_b, m, pos, geo["llm_heads"], geo["llm_kv_heads"], geo.get("rope_theta", 1e6), mode,
                                          return_kv=True)
            else:
                rows = (torch.arange(B)[:, None] * nb + parents).view(-1)          # the cache follows the beams
                cache = [(k[rows], v[rows]) for k, v in cache]
                pos = m.long().sum(-1, keepdim=True) - 1
                hid, cache = qwen2_hidden_step(W, table[toks[:, -1:]], cache, m, pos, geo["llm_heads"], geo["llm_kv_heads"],
                                               geo.get("rope_theta", 1e6), mode)
            logits = linear(hid[:, -1], lm_head_weight(W), None, mode).float()
        else:
            x = torch.cat([emb_b, table[toks]], 1)
            m = torch.cat([mask_b.bool(), torch.ones(B * nb, cur, dtype=torch.bool)], 1)
            pos = (m.long().cumsum(-1) - 1).masked_fill(~m, 1)
            hid = qwen2_hidden(W, x, m, pos, geo["llm_heads"], geo["llm_kv_heads"], geo.get("rope_theta", 1e6), mode)
            logits = linear(hid[:, -1], lm_head_weight(W), None, mode).float()
        if step_times is not None:
            import time
            step_times.append(time.perf_counter())
        if logits_trace is not None:
            logits_trace.append((logits.clone(), toks.clone()))
        if logit_jitter is not None:
            logits = logit_jitter(logits)
        logp = torch.log_softmax(logits, -1)
        if cur < min_length:
            logp[:, eos] = float("-inf")
        acc = (logp.view(B, nb, V) + run_scores[:, :, None]).view(B, nb * V)
        top_lp, top_ix = torch.topk(acc, K)
        beam_ix, tok = top_ix // V, top_ix % V
        cand = torch.gather(run_seq, 1, beam_ix[:, :, None].expand(-1, -1, max_new_tokens)).clone()
        cand[:, :, cur] = tok
        stop = (tok == eos) | (cur + 1 >= max_new_tokens)
        # running beams for the next step
        run_lp = top_lp + stop.float() * NEG
        nxt = torch.topk(run_lp, nb)[1]
        run_seq = torch.gather(cand, 1, nxt[:, :, None].expand(-1, -1, max_new_tokens))
        run_scores = torch.gather(run_lp, 1, nxt)
        parents = torch.gather(beam_ix, 1, nxt)                  # the beam each running hypothesis continues (kv_cache)
        # finished beams
        just = stop & top_mask[None]
        sc = top_lp / ((cur + 1) ** length_penalty)
        sc = sc + (~unsat).float() * NEG + (~just).float() * NEG
        m_seq = torch.cat([fin_seq, cand], 1)
        m_sc = torch.cat([fin_scores, sc], 1)
        m_len = torch.cat([fin_len, torch.full((B, K), cur + 1, dtype=torch.long)], 1)
        m_fin = torch.cat([is_fin, just], 1)
        keep = torch.topk(m_sc, nb)[1]
        fin_seq = torch.gather(m_seq, 1, keep[:, :, None].expand(-1, -1, max_new_tokens))
        fin_scores = torch.gather(m_sc, 1, keep)
        fin_len = torch.gather(m_len, 1, keep)
        is_fin = torch.gather(m_fin, 1, keep)
        cur += 1
        best_run = run_scores[:, :1] / (cur ** length_penalty)
        worst_fin = torch.where(is_fin, fin_scores.min(1, keepdim=True)[0], torch.full_like(fin_scores, NEG))
        unsat = unsat & (best_run > worst_fin).any(-1, keepdim=True)
        if not (bool(unsat.any()) and not bool(stop.all())):
            break
    n = int(fin_len[:, 0].max())
    return fin_seq[:, 0, :n]
