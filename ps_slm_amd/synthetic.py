"""Seeded synthetic weights and batches (there are no pretrained weights, tokenizer files or datasets on the
benchmark box).  Shapes follow SURVEY.md section 8d: a "30 s-equivalent utterance" = 25 prompt ids (one
<speech>), 104 pseudo-posterior tokens, 128 target ids (incl. EOS) -> merged S = 256.

``random_state_dict`` produces a REFERENCE-NAMED state dict (``llm.*``, ``encoder_projector.*``, ``encoder.*``)
so the same tensors can be loaded into the real reference (oracle/make_golden.py), the oracle and TasuModel.
"""
import math

import numpy as np
import torch

from .model import Geometry

MID_GEOMETRY = dict(llm_vocab=1000, llm_dim=256, llm_inter=512, llm_layers=2, llm_heads=2, llm_kv_heads=1,
                    rope_theta=1e6, tied=True, ctc_vocab=203, bottleneck=128, feat_dim=80, enc_dim=256, enc_heads=2,
                    enc_ffn=512, enc_blocks=2, enc_tp_blocks=1, enc_kernel=11, speech_id=990, eos_id=980)


def random_state_dict(geo: Geometry, seed: int, with_encoder=True, scale=0.05):
    """CPU fp32, deterministic in (geo, seed).  Norm weights are perturbed around 1 and biases are non-zero so
    that every parameter matters in parity tests."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def rn(*shape, s=scale):
        return torch.randn(*shape, generator=g) * s

    def near_one(n):
        return 1.0 + 0.1 * torch.randn(n, generator=g)

    D, I, H, G, V = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads, geo.llm_vocab
    hd = 128
    sd["llm.model.embed_tokens.weight"] = rn(V, D)
    for l in range(geo.llm_layers):
        p = f"llm.model.layers.{l}."
        sd[p + "input_layernorm.weight"] = near_one(D)
        sd[p + "post_attention_layernorm.weight"] = near_one(D)
        sd[p + "self_attn.q_proj.weight"] = rn(H * hd, D)
        sd[p + "self_attn.q_proj.bias"] = rn(H * hd)
        sd[p + "self_attn.k_proj.weight"] = rn(G * hd, D)
        sd[p + "self_attn.k_proj.bias"] = rn(G * hd)
        sd[p + "self_attn.v_proj.weight"] = rn(G * hd, D)
        sd[p + "self_attn.v_proj.bias"] = rn(G * hd)
        sd[p + "self_attn.o_proj.weight"] = rn(D, H * hd)
        sd[p + "mlp.gate_proj.weight"] = rn(I, D)
        sd[p + "mlp.up_proj.weight"] = rn(I, D)
        sd[p + "mlp.down_proj.weight"] = rn(D, I)
    sd["llm.model.norm.weight"] = near_one(D)
    if not geo.tied:
        sd["llm.lm_head.weight"] = rn(V, D)
    K, Hb = geo.pin, geo.bottleneck                     # projector input width: the CTC vocabulary, or enc_dim for raw features
    if geo.projector == "cross-attention":              # EncoderProjectorCTCCA: one matrix, W_q [llm_dim, K], no bias
        sd["encoder_projector.W_q.weight"] = rn(D, K, s=1.0 / math.sqrt(K))
    elif geo.projector == "cov1d-linear":                 # EncoderProjectorCov1d: Conv1d(K, K, k, stride k) -> ReLU -> Linear -> ReLU -> Linear
        kc = geo.projector_ds_rate
        sd["encoder_projector.conv1d.weight"] = rn(K, K, kc, s=1.0 / math.sqrt(K * kc))
        sd["encoder_projector.conv1d.bias"] = rn(K)
        sd["encoder_projector.linear1.weight"] = rn(Hb, K, s=1.0 / math.sqrt(K))
        sd["encoder_projector.linear1.bias"] = rn(Hb)
        sd["encoder_projector.linear2.weight"] = rn(D, Hb, s=1.0 / math.sqrt(Hb))
        sd["encoder_projector.linear2.bias"] = rn(D)
    elif geo.projector == "linear":                     # EncoderProjectorConcat: k frames concatenated, no norm
        Kin = K * geo.projector_ds_rate
        sd["encoder_projector.linear1.weight"] = rn(Hb, Kin, s=1.0 / math.sqrt(K))
        sd["encoder_projector.linear1.bias"] = rn(Hb)
        sd["encoder_projector.linear2.weight"] = rn(D, Hb, s=1.0 / math.sqrt(Hb))
        sd["encoder_projector.linear2.bias"] = rn(D)
    else:
        sd["encoder_projector.norm.weight"] = near_one(K)
        sd["encoder_projector.norm.bias"] = rn(K)
        sd["encoder_projector.ffn.0.weight"] = rn(Hb, K, s=1.0 / math.sqrt(K))
        sd["encoder_projector.ffn.0.bias"] = rn(Hb)
        sd["encoder_projector.ffn.2.weight"] = rn(D, Hb, s=1.0 / math.sqrt(Hb))
        sd["encoder_projector.ffn.2.bias"] = rn(D)
    if with_encoder:
        E, Fd, Ff, ks = geo.enc_dim, geo.feat_dim, geo.enc_ffn, geo.enc_kernel
        sd["encoder.embed.weight"] = rn(16, Fd, s=1.0)
        groups = [("encoders0", 1, Fd), ("encoders", geo.enc_blocks - 1, E), ("tp_encoders", geo.enc_tp_blocks, E)]
        for name, n, in_dim in groups:
            for i in range(n):
                p = f"encoder.encoder.{name}.{i}."
                sd[p + "norm1.weight"] = near_one(in_dim)
                sd[p + "norm1.bias"] = rn(in_dim)
                sd[p + "norm2.weight"] = near_one(E)
                sd[p + "norm2.bias"] = rn(E)
                sd[p + "self_attn.linear_q_k_v.weight"] = rn(3 * E, in_dim, s=1.0 / math.sqrt(in_dim))
                sd[p + "self_attn.linear_q_k_v.bias"] = rn(3 * E)
                sd[p + "self_attn.linear_out.weight"] = rn(E, E, s=1.0 / math.sqrt(E))
                sd[p + "self_attn.linear_out.bias"] = rn(E)
                sd[p + "self_attn.fsmn_block.weight"] = rn(E, 1, ks, s=0.2)
                sd[p + "feed_forward.w_1.weight"] = rn(Ff, E, s=1.0 / math.sqrt(E))
                sd[p + "feed_forward.w_1.bias"] = rn(Ff)
                sd[p + "feed_forward.w_2.weight"] = rn(E, Ff, s=1.0 / math.sqrt(Ff))
                sd[p + "feed_forward.w_2.bias"] = rn(E)
        for nm in ("after_norm", "tp_norm"):
            sd[f"encoder.encoder.{nm}.weight"] = near_one(E)
            sd[f"encoder.encoder.{nm}.bias"] = rn(E)
        sd["encoder.ctc.ctc_lo.weight"] = rn(geo.ctc_vocab, E, s=1.0 / math.sqrt(E))
        sd["encoder.ctc.ctc_lo.bias"] = rn(geo.ctc_vocab)
    return sd


def decode_fixture_state_dict(geo: Geometry, seed: int, eos_boost=2.5):
    """Weights of the decode fixtures whose token ids are compared exactly (oracle/make_golden_generate_margin.py,
    tests/golden/mid_generate_margin.npz): ``random_state_dict`` without the encoder, with the EOS row of the embedding /
    lm_head table scaled up so that beams do finish inside a dozen positions (a random-init head almost never emits EOS)."""
    sd = random_state_dict(geo, seed, with_encoder=False)
    for key in ("llm.model.embed_tokens.weight", "llm.lm_head.weight"):
        if key in sd:
            sd[key][geo.eos_id] *= eos_boost
    return sd


def synthetic_text_batch(geo: Geometry, B, seed, prompt_len=25, n_audio=104, target_len=128, speech_pos=12,
                         feat_frames=500, noise=True, drop_prob=0.0, ragged=False):
    """One fixed-length (or ragged, for tests) text-only batch in the collator's schema
    (Multitask/dataset/speech_dataset_large.py:290-305) plus the explicit CPS draws (ids, alpha, keep)."""
    g = torch.Generator().manual_seed(seed)
    hi = min(geo.eos_id, geo.speech_id)
    rows, post_ids, alphas, keeps = [], [], [], []
    for b in range(B):
        pl = prompt_len - (b % 3 if ragged else 0)
        tl = target_len - (2 * b % 5 if ragged else 0)
        na = n_audio - (3 * b % 7 if ragged else 0)
        prompt = torch.randint(0, hi, (pl,), generator=g).tolist()
        prompt[min(speech_pos, pl - 1)] = geo.speech_id
        target = torch.randint(0, hi, (tl - 1,), generator=g).tolist() + [geo.eos_id]
        rows.append((prompt, target))
        post_ids.append(torch.randint(1, geo.ctc_vocab, (na,), generator=g).tolist())
        alphas.append(float(torch.empty(()).uniform_(0.0, 0.1, generator=g)) if noise else 0.0)
        keeps.append((torch.rand(na, generator=g) >= drop_prob).numpy())
    L = max(len(p) + len(t) for p, t in rows)
    ids = np.full((B, L), geo.eos_id, dtype=np.int64)
    am = np.zeros((B, L), dtype=bool)
    lab = np.full((B, L), -100, dtype=np.int64)
    for b, (p, t) in enumerate(rows):
        n = len(p) + len(t)
        ids[b, :n] = p + t
        am[b, :n] = True
        lab[b, len(p):n] = t
    feats = torch.randn(B, feat_frames, geo.feat_dim, generator=g).half().float()
    batch = dict(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(am), labels=torch.from_numpy(lab),
                 input_features=feats, input_feature_length=torch.full((B,), feat_frames, dtype=torch.long),
                 post_ids=post_ids)
    if noise:
        batch["alphas"] = alphas
        batch["keeps"] = keeps
    return batch


def random_lora_state_dict(geo: Geometry, cfg, seed: int, b_scale=0.05):
    """Seeded adapter tensors under the reference's checkpoint keys (ps_slm_amd.lora.key_of): A ~ U(-1/sqrt(in), 1/sqrt(in)) as
    peft initialises it, B ~ N(0, b_scale) -- NOT peft's zero init, so that every adapter contributes to the outputs and
    every lora_A receives a gradient in parity tests."""
    from .lora import key_of, target_dims
    g = torch.Generator().manual_seed(seed)
    dims = target_dims(geo)
    sd = {}
    for l in range(geo.llm_layers):
        for t in cfg.target_modules:
            i, o = dims[t]
            sd[key_of(l, t, "A")] = (torch.rand(cfg.r, i, generator=g) * 2 - 1) / math.sqrt(i)
            sd[key_of(l, t, "B")] = torch.randn(o, cfg.r, generator=g) * b_scale
    return sd
