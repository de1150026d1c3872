"""Device-resident TASU model: weights laid out for the gfx950 kernels + the hand-scheduled forward/backward of
the alignment training step (no autograd tape, no tracing compiler: the graph is fixed because the LLM and the
encoder are frozen -- Multitask/scripts/finetune_deespeed_sensevoice.sh:44,84 -- so backward is dgrad-only
through the decoder and wgrad only for the 6 projector tensors).

Reference arithmetic being replaced:
  slam_model_asr.forward            Multitask/model/ps-slm.py:411-537
  EncoderProjectorLinearSiLU        Multitask/model/projector.py:129-151
  Qwen2ForCausalLM.forward / loss   transformers modeling_qwen2.py:423-470, loss/loss_utils.py:49-71
  token accuracy                    Multitask/utils/metric.py:3-20

Numerics ("bf16" mode = torch.autocast(bfloat16) semantics of Multitask/utils/deepspeed_utils.py:160): bf16 GEMM
operands/results with fp32 MFMA accumulation, fp32 residual stream, fp32 norms / softmax / loss, fp32 master
projector weights and optimizer state.

HBM layout (sized for 288 GB: everything stays resident, nothing is recomputed):
  * every frozen Linear weight twice, bf16: W [out,in] for forward and W^T [in,out] for dgrad, so that one
    K-contiguous "NT" MFMA kernel serves both;  q/k/v and gate/up are fused along N.
  * embedding table fp32 (gather) + bf16 lm_head copies [V,D] and [D,Vpad].
  * projector parameters in ONE flat fp32 buffer (+ flat grad / m / v / bf16 working copy); the 25055-wide K
    dimension is padded to a multiple of 64 with zeros (pad columns provably stay zero under AdamW).
  * per layer saved for backward: x_in, x_mid (fp32), rstd1/2, rotated qkv, attention out, lse, gate|up.
"""
import collections
import math
import os
from dataclasses import dataclass, field

import numpy as np
import torch

from .merge import build_merge_plan
from .ops import GEMM_BF16, GEMM_F32, GEMM_RESID, LN_BWD_SPLIT

HD = 128


def rup(x, m):
    return (x + m - 1) // m * m


@dataclass
class Geometry:
    # Qwen2.5 decoder
    llm_vocab: int = 151936
    llm_dim: int = 1536
    llm_inter: int = 8960
    llm_layers: int = 28
    llm_heads: int = 12
    llm_kv_heads: int = 2
    rope_theta: float = 1e6
    rms_eps: float = 1e-6
    tied: bool = True
    # projector
    ctc_vocab: int = 25055
    bottleneck: int = 2048
    ln_eps: float = 1e-5
    projector: str = "linear-silu"   # "linear-silu" (EncoderProjectorLinearSiLU, projector.py:129-151: the shipped recipe) or
                                     # "linear" (EncoderProjectorConcat, projector.py:28-49: k frames concatenated, ReLU, no norm)
    projector_ds_rate: int = 1       # k of the "linear" / "cov1d-linear" projectors (model_config.encoder_projector_ds_rate)
    ca_heads: int = 8                # heads of the "cross-attention" projector (EncoderProjectorCTCCA, projector.py:105: n_heads=8)
    proj_in: int = 0                 # width of a projector input frame; 0 = ctc_vocab (the CTC posterior, train_config.ctc_posterior=true);
                                     # enc_dim for the raw-feature branch (ctc_posterior=false, ps-slm.py:515-523: encoder states)
    # SenseVoiceSmall encoder
    feat_dim: int = 560
    enc_dim: int = 512
    enc_heads: int = 4
    enc_ffn: int = 2048
    enc_blocks: int = 50
    enc_tp_blocks: int = 20
    enc_kernel: int = 11
    blank_id: int = 0
    # tokens
    speech_id: int = 151665
    eos_id: int = 151643

    @property
    def pin(self):
        return self.proj_in or self.ctc_vocab

    @classmethod
    def qwen25_1p5b(cls):
        return cls()

    @classmethod
    def qwen25_7b(cls):
        return cls(llm_vocab=152064, llm_dim=3584, llm_inter=18944, llm_layers=28, llm_heads=28, llm_kv_heads=4,
                   tied=False)

    @classmethod
    def from_dict(cls, d):
        names = {f for f in cls.__dataclass_fields__}
        return cls(**{k: v for k, v in d.items() if k in names})


PROJ_NAMES = ("norm.weight", "norm.bias", "ffn.0.weight", "ffn.0.bias", "ffn.2.weight", "ffn.2.bias")
PROJ_NAMES_LINEAR = ("linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias")
PROJ_NAMES_COV1D = ("conv1d.weight", "conv1d.bias") + PROJ_NAMES_LINEAR
PROJ_NAMES_CA = ("W_q.weight",)


class ProjectorParams:
    """The only trainable tensors (54,512,062 parameters at full geometry) in one flat, padded buffer.  Three projector kinds
    share the layout [optional input stage | W1 | b1 | W2 | b2]: the first Linear reads ``kin`` frames of the CTC vocabulary,
    each frame's K columns padded to Kp = a multiple of 64 (the pad columns of W1 stay zero: their inputs are zero).  Input
    stage: LayerNorm ("linear-silu"), nothing ("linear": kin = k frames concatenated), or ("cov1d-linear",
    EncoderProjectorCov1d, projector.py:53-73) Conv1d(K, K, kernel = stride = k) + ReLU, stored as the [Kp, k * Kp] matrix of
    the GEMM over k concatenated frames (W0[o, j * Kp + i] = conv.weight[o, i, j]) -- then kin = 1."""

    def __init__(self, geo: Geometry, device):
        K, Kp, Hb, Do = geo.pin, rup(geo.pin, 64), geo.bottleneck, geo.llm_dim
        self.kind = geo.projector
        if self.kind not in ("linear-silu", "linear", "cov1d-linear", "cross-attention"):
            raise NotImplementedError(f"encoder_projector {self.kind!r}")
        self.is_ca = self.kind == "cross-attention"
        if self.is_ca:
            # EncoderProjectorCTCCA (projector.py:104-126): ONE trainable matrix, W_q [llm_dim, K] (no bias); keys and values are
            # the LLM's embedding table.  Laid out like a first Linear with kin = 1 whose "bottleneck" is llm_dim.
            self.k = self.kin = 1
            self.has_norm = self.has_conv = False
            self.K, self.Kp, self.Hb, self.Do = K, Kp, Do, Do
            self.names = PROJ_NAMES_CA
            self.n_w1, self.n_b1, self.n_w2, self.n_b2 = "W_q.weight", None, None, None
            self.real = {"W_q.weight": (Do, K)}
            self.offsets = {"W_q.weight": (0, (Do, Kp))}
            self.numel = off = rup(Do * Kp, 64)
            f32 = dict(dtype=torch.float32, device=device)
            self.p, self.g, self.m, self.v = (torch.zeros(off, **f32) for _ in range(4))
            self.pb = torch.zeros(off, dtype=torch.bfloat16, device=device)
            self.w1b_t = self.w2b_t = None
            return
        self.k = int(geo.projector_ds_rate) if self.kind in ("linear", "cov1d-linear") else 1   # frames per projector row
        self.has_norm = self.kind == "linear-silu"
        self.has_conv = self.kind == "cov1d-linear"
        self.kin = 1 if self.has_conv else self.k           # frames the first Linear reads
        kc, k = self.k, self.kin
        self.K, self.Kp, self.Hb, self.Do = K, Kp, Hb, Do
        if self.has_norm:
            self.names = PROJ_NAMES
            self.n_w1, self.n_b1, self.n_w2, self.n_b2 = "ffn.0.weight", "ffn.0.bias", "ffn.2.weight", "ffn.2.bias"
            shapes = {"norm.weight": (Kp,), "norm.bias": (Kp,)}
            self.real = {"norm.weight": (K,), "norm.bias": (K,)}
        else:
            self.names = PROJ_NAMES_COV1D if self.has_conv else PROJ_NAMES_LINEAR
            self.n_w1, self.n_b1, self.n_w2, self.n_b2 = PROJ_NAMES_LINEAR
            shapes, self.real = {}, {}
            if self.has_conv:
                shapes = {"conv1d.weight": (Kp, kc * Kp), "conv1d.bias": (Kp,)}
                self.real = {"conv1d.weight": (K, K, kc), "conv1d.bias": (K,)}
        shapes.update({self.n_w1: (Hb, k * Kp), self.n_b1: (Hb,), self.n_w2: (Do, Hb), self.n_b2: (Do,)})
        self.real.update({self.n_w1: (Hb, k * K), self.n_b1: (Hb,), self.n_w2: (Do, Hb), self.n_b2: (Do,)})
        self.offsets, off = {}, 0
        for n in self.names:
            self.offsets[n] = (off, shapes[n])
            off += rup(int(np.prod(shapes[n])), 64)
        self.numel = off
        f32 = dict(dtype=torch.float32, device=device)
        self.p = torch.zeros(off, **f32)
        self.g = torch.zeros(off, **f32)
        self.m = torch.zeros(off, **f32)
        self.v = torch.zeros(off, **f32)
        self.pb = torch.zeros(off, dtype=torch.bfloat16, device=device)
        self.w1b_t = torch.zeros(k * Kp, Hb, dtype=torch.bfloat16, device=device)   # W1^T for dgrad (linear-silu, cov1d-linear)
        self.w2b_t = torch.zeros(Hb, Do, dtype=torch.bfloat16, device=device)       # W2^T for dgrad

    def extend(self, extra):
        """Grows the flat buffers by ``extra`` elements behind the projector's tensors (the LoRA adapters: ps_slm_amd/lora.py);
        contents are kept.  Everything that walks [0, numel) -- AdamW, the gradient exchange -- then covers the tail too."""
        n = self.numel + int(extra)
        for name in ("p", "g", "m", "v", "pb"):
            old = getattr(self, name)
            new = torch.zeros(n, dtype=old.dtype, device=old.device)
            new[: old.numel()].copy_(old)
            setattr(self, name, new)
        self.numel = n

    def view(self, flat, name):
        off, shp = self.offsets[name]
        return flat[off:off + int(np.prod(shp))].view(*shp)

    def num_parameters(self):
        return sum(int(np.prod(s)) for s in self.real.values())

    def _blocks(self, t2d):
        """[rows, kin * Kp] padded view -> [rows, kin, Kp] (frame blocks)."""
        return t2d.view(t2d.shape[0], self.kin, self.Kp)

    def _conv_blocks(self, t2d):
        """conv1d.weight as stored, [Kp, k * Kp] -> [Kp, k, Kp] (output channel, tap, input channel)."""
        return t2d.view(self.Kp, self.k, self.Kp)

    def load(self, name, t):
        """t: fp32 tensor with the REFERENCE shape (unpadded)."""
        dst = self.view(self.p, name)
        dst.zero_()
        if dst.dim() == 1:
            dst[: t.shape[0]].copy_(t)
        elif name == "conv1d.weight":                      # reference layout [out, in, tap]
            self._conv_blocks(dst)[: self.K, :, : self.K].copy_(t.permute(0, 2, 1))
        elif name == self.n_w1:
            self._blocks(dst)[:, :, : self.K].copy_(t.reshape(t.shape[0], self.kin, self.K))
        else:
            dst[:, : t.shape[1]].copy_(t)

    def _export(self, flat, name):
        src = self.view(flat, name)
        r = self.real[name]
        if src.dim() == 1:
            return src[: r[0]].detach().clone()
        if name == "conv1d.weight":
            return self._conv_blocks(src)[: self.K, :, : self.K].permute(0, 2, 1).contiguous().detach().clone()
        if name == self.n_w1:
            return self._blocks(src)[:, :, : self.K].reshape(r[0], r[1]).detach().clone()
        return src[:, : r[1]].detach().clone()

    def export(self, name):
        return self._export(self.p, name)

    def export_grad(self, name):
        return self._export(self.g, name)

    def refresh_working_copies(self, ops):
        """bf16 working copy (skipped when AdamW already wrote it) + the transposed copies for dgrad."""
        if self.is_ca:
            return                                         # nothing upstream of W_q needs a gradient
        kKp = self.kin * self.Kp
        if self.has_norm or self.has_conv:   # the "linear" projector's input carries no parameters: its W1^T is never needed
            ops.transpose(self.view(self.pb, self.n_w1), self.w1b_t, self.Hb, kKp, self.Hb, kKp)
        ops.transpose(self.view(self.pb, self.n_w2), self.w2b_t, self.Do, self.Hb, self.Do, self.Hb)


class LLMWeights:
    """Frozen Qwen2 weights, bf16, fused and doubly laid out (see module docstring)."""

    def __init__(self, geo: Geometry, device):
        self.geo = geo
        self.device = device
        self.layers = []
        self.embed = None      # fp32 [V, D]
        self.head = None       # bf16 [V, D]
        self.head_t = None     # bf16 [D, Vpad]
        self.norm = None       # fp32 [D]
        self._decode_ready = False
        self._stale_ptrs = []  # row-major weight addresses whose fragment-order copies (ops._frag) belong to replaced weights
        # fp32 arithmetic mode of the decode path (train_config.use_fp16 = false, ps_slm_amd/decode_fp32.py): fp32 copies of the
        # fused weights next to the bf16 ones, {"layers": [{wqkv, bqkv, wo, wgu, wd}], "head"}; set keep_f32 BEFORE loading
        self.keep_f32 = False
        self.f32 = None

    def _drop_weights(self):
        """Before a reload: remember which registered decode copies die with the old tensors."""
        for w in self.layers:
            self._stale_ptrs += [w[k].data_ptr() for k in ("wqkv", "wo", "wgu", "wd")]
        if self.head is not None:
            self._stale_ptrs.append(self.head.data_ptr())
        self.layers = []
        self.f32 = {"layers": [], "head": None} if self.keep_f32 else None
        self._decode_ready = False

    @staticmethod
    def _pair(w, device):
        """fp32 [N,K] -> (bf16 [N,K], bf16 [K,N]) on device.  Load-time layout work (not on the step path)."""
        wb = w.to(device=device, dtype=torch.bfloat16).contiguous()
        return wb, wb.t().contiguous()

    def add_layer(self, ln1, wq, bq, wk, bk, wv, bv, wo, ln2, wg, wu, wd):
        dev = self.device
        wqkv, wqkv_t = self._pair(torch.cat([wq, wk, wv], 0), dev)
        wo_b, wo_t = self._pair(wo, dev)
        wgu, wgu_t = self._pair(torch.cat([wg, wu], 0), dev)
        wd_b, wd_t = self._pair(wd, dev)
        self.layers.append(dict(
            ln1=ln1.to(dev, torch.float32).contiguous(), ln2=ln2.to(dev, torch.float32).contiguous(),
            wqkv=wqkv, wqkv_t=wqkv_t, bqkv=torch.cat([bq, bk, bv], 0).to(dev, torch.bfloat16).contiguous(),
            wo=wo_b, wo_t=wo_t, wgu=wgu, wgu_t=wgu_t, wd=wd_b, wd_t=wd_t))
        if self.keep_f32:
            if self.f32 is None:
                self.f32 = {"layers": [], "head": None}
            c = lambda t: t.to(dev, torch.float32).contiguous()
            self.f32["layers"].append(dict(wqkv=c(torch.cat([wq, wk, wv], 0)), bqkv=c(torch.cat([bq, bk, bv], 0)), wo=c(wo),
                                           wgu=c(torch.cat([wg, wu], 0)), wd=c(wd)))

    def set_embed(self, embed, head, norm):
        geo, dev = self.geo, self.device
        V, D = geo.llm_vocab, geo.llm_dim
        Vp = rup(V, 64)
        self.embed = embed.to(dev, torch.float32).contiguous()
        hb = head.to(dev, torch.bfloat16).contiguous()
        self.head = hb
        self.head_t = torch.zeros(D, Vp, dtype=torch.bfloat16, device=dev)
        self.head_t[:, :V].copy_(hb.t())
        self.norm = norm.to(dev, torch.float32).contiguous()
        if self.keep_f32:
            if self.f32 is None:
                self.f32 = {"layers": [], "head": None}
            self.f32["head"] = self.embed if head is embed else head.to(dev, torch.float32).contiguous()   # tied: one copy

    def prepare_decode(self, ops):
        """Decode-step copies of the frozen weights in the order the weight-streaming kernels consume them
        (ops.register_decode_weight; once per model, +1x the decoder's bf16 bytes)."""
        if self._decode_ready:
            return
        geo = self.geo
        if self._stale_ptrs and hasattr(ops, "forget_decode_weights"):
            ops.forget_decode_weights(self._stale_ptrs)           # copies of the weights this model held before a reload
        self._stale_ptrs = []
        self._decode_ready = True
        # fragment-order copies are read only when the whole layer runs in fragment order (ops.begin_decode): at Qwen2.5-7B
        # (D = 3584 streams as 7 x 512) it does not, and ~10 GB of copies would never be used
        if hasattr(ops, "decode_frag_possible") and not ops.decode_frag_possible(geo.llm_dim, geo.llm_heads * HD):
            return
        for w in self.layers:
            ops.register_decode_weight(w["wqkv"], "qkv", w["wqkv"].shape[0], geo.llm_heads, geo.llm_kv_heads)
            ops.register_decode_weight(w["wo"], "plain", w["wo"].shape[0])
            ops.register_decode_weight(w["wgu"], "swiglu", geo.llm_inter)
            ops.register_decode_weight(w["wd"], "plain", w["wd"].shape[0], slabs_ok=True)
        ops.register_decode_weight(self.head, "plain", self.head.shape[0])

    def load_reference_state_dict(self, sd, pre="llm."):
        geo = self.geo
        self._drop_weights()
        for l in range(geo.llm_layers):
            p = f"{pre}model.layers.{l}."
            self.add_layer(sd[p + "input_layernorm.weight"],
                           sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"],
                           sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"],
                           sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"],
                           sd[p + "self_attn.o_proj.weight"], sd[p + "post_attention_layernorm.weight"],
                           sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"], sd[p + "mlp.down_proj.weight"])
        emb = sd[pre + "model.embed_tokens.weight"]
        head = sd.get(pre + "lm_head.weight", emb)
        self.set_embed(emb, head, sd[pre + "model.norm.weight"])

    def init_random(self, seed):
        """Seeded N(0, 0.02) linears/embedding, ones norms (HF default init), generated ON DEVICE tensor by tensor
        (no pretrained weights exist on the benchmark box)."""
        geo, dev = self.geo, self.device
        self._drop_weights()
        g = torch.Generator(device=dev).manual_seed(seed)
        D, I, H, G = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads

        def rn(*shape):
            return torch.randn(*shape, generator=g, device=dev, dtype=torch.float32) * 0.02

        ones = torch.ones(D, device=dev)
        for _ in range(geo.llm_layers):
            self.add_layer(ones, rn(H * HD, D), rn(H * HD), rn(G * HD, D), rn(G * HD), rn(G * HD, D), rn(G * HD),
                           rn(D, H * HD), ones, rn(I, D), rn(I, D), rn(D, I))
        emb = rn(geo.llm_vocab, D)
        self.set_embed(emb, emb if geo.tied else rn(geo.llm_vocab, D), ones)


class UploadPack:
    """The step's small host -> device inputs (merge plan, labels, masks, pseudo-posterior ids: 14 arrays, < 1 MB) as ONE
    asynchronous copy.  Every name owns a fixed slot of one device buffer (stable addresses: captured hipGraphs keep reading them);
    ``put`` writes the array into the same offset of a PINNED host buffer, ``flush`` issues one H2D copy per run of adjacent slots
    on the current stream, with no host synchronisation (a copy out of pageable memory makes the host wait for the stream to
    drain: the host could never run ahead of the device, e.g. to collate the next batch).  MEASURED: under rocprofv3 the 14
    per-array copies showed up as 0.48 ms of idle GPU per 28.4-ms step (tools/trace_gaps.py), but that is the profiler's own
    serialisation -- without it the step time is the same either way (alternating processes on one box, tools/ab_env.sh
    TASU_UPLOAD_PACK 0 1: 28.12 / 28.06 / 28.14 against 28.07 / 28.18 / 28.11 ms).  Kept for the host-side property only.
    The pinned buffers form a ring: a buffer is rewritten only after the copy that last read it has completed."""
    RING = 4

    def __init__(self, model, capacity=32 << 20):
        self.model, self.cap = model, capacity
        self.dev = torch.empty(capacity, dtype=torch.uint8, device=model.device)
        self.host = [torch.empty(capacity, dtype=torch.uint8).pin_memory() for _ in range(self.RING)]
        self.done = [None] * self.RING
        self.cur, self.open, self.end = 0, False, 0
        self.slots = {}                                  # name -> (offset, capacity in bytes)
        self.pending = []                                # (offset, bytes, slot capacity) since the last flush

    def put(self, name, arr):
        arr = np.ascontiguousarray(arr)
        t = torch.from_numpy(arr)
        nb = arr.nbytes
        if nb == 0:
            return torch.empty(arr.shape, dtype=t.dtype, device=self.model.device)
        slot = self.slots.get(name)
        if slot is None or slot[1] < nb:
            cap = rup(nb if slot is None else 2 * nb, 256)
            if self.end + cap > self.cap:
                raise RuntimeError(f"UploadPack: {self.end + cap} bytes of step inputs exceed the {self.cap}-byte pack")
            if slot is not None:
                self.model._buf_gen += 1                 # the name moved: graphs captured on the old address are dead
            slot = self.slots[name] = (self.end, cap)
            self.end += cap
        if not self.open:                                # first array since the last flush: the next pinned buffer of the ring
            self.cur = (self.cur + 1) % self.RING
            if self.done[self.cur] is not None:
                self.done[self.cur].synchronize()        # (four flushes ago: long complete)
            self.open = True
        off = slot[0]
        self.host[self.cur][off:off + nb].numpy()[:] = arr.reshape(-1).view(np.uint8)
        self.pending.append((off, nb, slot[1]))
        return self.dev[off:off + nb].view(t.dtype).view(arr.shape)

    def flush(self):
        if not self.pending:
            return
        runs = []
        for off, nb, cap in sorted(self.pending):
            if runs and runs[-1][2] == off:              # begins where the previous array's SLOT ends: one copy covers both
                runs[-1] = (runs[-1][0], off + nb, off + cap)
            else:
                runs.append((off, off + nb, off + cap))
        h = self.host[self.cur]
        for lo, hi, _ in runs:
            self.dev[lo:hi].copy_(h[lo:hi], non_blocking=True)
        ev = self.done[self.cur] or torch.cuda.Event()
        ev.record()
        self.done[self.cur] = ev
        self.pending, self.open = [], False


@dataclass
class StepState:
    """Everything one training step needs between forward and backward (device tensors + the host plan)."""
    plan: object = None
    B: int = 0
    S: int = 0
    M: int = 0
    Ra: int = 0        # projector rows (B * Lmax)
    Rap: int = 0       # padded to 64
    Fap: int = 0       # posterior FRAME rows, padded (= Rap * k: the "linear" projector concatenates k frames per row)
    nL: int = 0        # positions that carry a label (shift_labels >= 0)
    nLp: int = 0       # padded to 64
    dev: dict = field(default_factory=dict)
    out: object = None
    path: str = "text"   # "text" (pseudo-posterior) or "audio" (encoder + PSD): part of the graph keys, the buffers differ
    lora_drop: bool = False   # this step's forward drew LoRA dropout masks (the backward regenerates them)
    backward_ran: bool = False   # the step's backward has run (TasuEngine.backward or outputs.loss.backward(): ps_slm._HipStep)
    fp32: bool = False           # the forward ran on the fp32 training path (train_fp32.forward_train_fp32): so must the backward


class TasuModel:
    def __init__(self, geo: Geometry, ops, device, keep_logits=True):
        self.geo, self.ops, self.device = geo, ops, torch.device(device)
        self.llm = LLMWeights(geo, self.device)
        self.proj = ProjectorParams(geo, self.device)
        self.encoder = None            # ps_slm_amd.encoder.EncoderWeights (audio path)
        self.keep_logits = keep_logits
        self._ws = {}
        self._pack = None               # UploadPack: the step's host -> device inputs (created on the first upload on a GPU)
        self.arith = "bf16"             # arithmetic of generate(): "bf16" (autocast semantics) or "fp32" (ps_slm_amd/decode_fp32.py)
        self.training = True
        # hipGraph replay of the (shape-static) forward / backward launch sequences: ~700 launches per step collapse
        # into two graph launches.  Keyed by the shapes baked into kernel arguments; the first call of a key runs
        # eagerly (allocates the workspace), the second is captured, later ones replay.
        self.use_graphs = False
        self.lora_bwd_graphs = os.environ.get("TASU_LORA_BWD_GRAPH", "0") == "1"      # (A/B: the adapted backward as a hipGraph; see run_backward)
        # (token-column multiple, posterior-row multiple, labelled-row multiple) or None.  Real data gives every batch its own shape
        # (CPS drops, frame-budget batching), and eagerly launched steps run ~10 % slower than graph replay (measured: 34.2 vs
        # 30.8 ms).  With buckets the training batch is padded to the next multiple -- masked token columns, zero posterior rows,
        # ignored label rows: exactly the padding the collator itself produces for a longer batch, so loss and gradients do not
        # change -- and the step's graphs are kept in a small LRU, so that shapes recur.
        self.shape_buckets = None
        # training step: the LAST decoder layer's MLP (and everything after it) runs on the labelled rows only -- the hidden state
        # of a position without a label feeds nothing after that layer's attention (its K / V, which all rows still produce) --
        # like the loss head (_loss_on_labelled_rows).  Same loss and gradients (tests compare both settings of the attribute).
        self.tail_rows = True
        self.graph_cache_size = 64
        self.decode_graphs = True      # the decode step (ps_slm_amd/decode.py) is always replayed as a graph on the GPU
        self._graphs = {}
        self._graph_seen = {}
        self._buf_gen = 0              # bumped whenever a named workspace buffer is re-allocated (grown)
        self._dec_graphs, self._dec_seen = collections.OrderedDict(), {}   # decode-step graphs (ps_slm_amd/decode.py): small LRU
        self._done_host = None         # pinned "decode finished" word the beam-update kernel writes
        self.lora = None               # ps_slm_amd.lora.LoraParams once enable_lora() ran (use_peft=true)
        self.freeze_projector = False  # train_config.freeze_projector (ps-slm.py:50-54): the projector's weight gradients, exchange and
                                       # optimizer update are skipped; only the adapters train (the plugin refuses it without use_peft)
        self.raw_features = geo.proj_in not in (0, geo.ctc_vocab)   # ctc_posterior=false: the projector reads encoder states
        self._lora_run = None
        # the frozen encoder one batch ahead (prefetch_encoder): the pending pass, the side stream it runs on, the event behind
        # the last PSD (which reads the encoder's output buffers), and the (B, T) shapes whose workspace exists at _buf_gen
        self._enc_ahead = None
        self._enc_stream = None
        self._psd_done = None
        self._enc_shapes = {}

    # ------------------------------------------------------------------------------------------ weights
    def load_reference_state_dict(self, sd):
        """sd: reference-named tensors (``llm.*``, ``encoder_projector.*``, optionally ``encoder.*``)."""
        self.llm.load_reference_state_dict(sd)
        if self.lora is not None:
            self.lora.build_ext(self.llm)                  # [W | B] copies of the adapted Linears follow the new base weights
        for n in self.proj.names:
            self.proj.load(n, sd["encoder_projector." + n].to(self.device, torch.float32))
        self.sync_projector_copies()
        if any(k.startswith("encoder.") for k in sd):
            from .encoder import EncoderWeights
            self._encoder_replaced()
            self.encoder = EncoderWeights(self.geo, self.device, keep_f32=self.llm.keep_f32)
            self.encoder.load_reference_state_dict(sd)

    def _encoder_replaced(self):
        """New encoder weights are about to be installed: a prefetched pass and the captured encoder graphs refer to the old
        tensors."""
        if self._enc_ahead is not None:
            self._enc_ahead["ready"].synchronize()
            self._enc_ahead = None
        for k in [k for k in self._graphs if k[:2] == ("region", "encoder")]:
            del self._graphs[k]
            self._graph_seen.pop(k, None)

    def init_random(self, seed=1234, with_encoder=False):
        self.llm.init_random(seed)
        if self.lora is not None:
            self.lora.build_ext(self.llm)
        self.init_projector_default(seed + 1)
        if with_encoder:
            from .encoder import EncoderWeights
            self._encoder_replaced()
            self.encoder = EncoderWeights(self.geo, self.device, keep_f32=self.llm.keep_f32)
            self.encoder.init_random(seed + 2)

    def init_projector_default(self, seed=42):
        """Default nn.Module init of the projector (EncoderProjectorLinearSiLU, projector.py:137-147, which zeroes its last
        bias; EncoderProjectorConcat, :28-37, plain nn.Linear init) when training starts from pretrained LLM/encoder weights and
        no projector checkpoint."""
        geo, dev, pr = self.geo, self.device, self.proj
        g = torch.Generator(device=dev).manual_seed(seed)
        K, Hb, Do = geo.pin * pr.kin, geo.bottleneck, geo.llm_dim

        def un(shape, fan_in):
            b = 1.0 / math.sqrt(fan_in)
            return (torch.rand(*shape, generator=g, device=dev, dtype=torch.float32) * 2 - 1) * b

        if pr.is_ca:
            pr.load("W_q.weight", un((Do, K), K))
            self.sync_projector_copies()
            return
        if pr.has_norm:
            pr.load("norm.weight", torch.ones(K, device=dev))
            pr.load("norm.bias", torch.zeros(K, device=dev))
        if pr.has_conv:                                    # nn.Conv1d default init: fan_in = in_channels * kernel_size
            Kc = geo.pin
            pr.load("conv1d.weight", un((Kc, Kc, pr.k), Kc * pr.k))
            pr.load("conv1d.bias", un((Kc,), Kc * pr.k))
        pr.load(pr.n_w1, un((Hb, K), K))
        pr.load(pr.n_b1, un((Hb,), K))
        pr.load(pr.n_w2, un((Do, Hb), Hb))
        pr.load(pr.n_b2, torch.zeros(Do, device=dev) if pr.has_norm else un((Do,), Hb))
        self.sync_projector_copies()

    def load_encoder_checkpoint(self, path):
        """funasr SenseVoiceSmall ``model.pt`` (keys ``encoder.*``, ``ctc.ctc_lo.*``, ``embed.weight``)."""
        from .encoder import EncoderWeights
        raw = torch.load(path, map_location="cpu")
        raw = raw.get("state_dict", raw)
        sd = {"encoder." + k: v.float() for k, v in raw.items()}
        self._encoder_replaced()
        self.encoder = EncoderWeights(self.geo, self.device, keep_f32=self.llm.keep_f32)
        self.encoder.load_reference_state_dict(sd)

    def sync_projector_copies(self):
        self.ops.cast_bf16(self.proj.p, self.proj.pb)
        self.refresh_working_copies()

    def refresh_working_copies(self):
        """After the bucket's bf16 copy changed (a load, an optimizer step): the transposed copies the dgrads read."""
        self.proj.refresh_working_copies(self.ops)
        if self.lora is not None:
            self.lora.refresh_working_copies(self.ops)

    def enable_lora(self, cfg, seed=4242):
        """use_peft=true (ps-slm.py:114-117): adapters on the decoder's Linears, trainable next to the projector.  Call before
        an engine is built on the model (the trainable bucket grows)."""
        from .lora import LoraParams, LoraRunner
        if self.lora is not None:
            raise RuntimeError("LoRA is already enabled on this model")
        self.lora = LoraParams(self.geo, cfg, self.proj, self.device)
        self.lora.build_ext(self.llm)
        self.lora.init_default(seed)
        self.lora.seed_dropout(seed)
        self._lora_run = LoraRunner(self)
        self._graphs, self._graph_seen = {}, {}
        self.sync_projector_copies()

    def lora_spans(self, n=7):
        """The decoder's layers as at most ``n`` spans (hi, lo), last layers first: the units of the adapters' gradient exchange."""
        L = self.geo.llm_layers
        n = max(1, min(n, L))
        cuts = [L - (L * i) // n for i in range(n + 1)]
        return [(cuts[i], cuts[i + 1]) for i in range(n) if cuts[i] > cuts[i + 1]]

    def lora_state_dict(self):
        return {} if self.lora is None else self.lora.state_dict()

    def lora_grads(self):
        return {} if self.lora is None else self.lora.grads()

    def projector_state_dict(self):
        return {"encoder_projector." + n: self.proj.export(n) for n in self.proj.names}

    def projector_grads(self):
        return {"encoder_projector." + n: self.proj.export_grad(n) for n in self.proj.names}

    # ------------------------------------------------------------------------------------------ workspace
    def _buf(self, name, shape, dtype):
        """Grow-only named buffers: allocated once per capacity, then reused every step."""
        n = int(np.prod(shape))
        t = self._ws.get(name)
        if t is None or t.numel() < n or t.dtype != dtype:
            if t is not None:
                self._buf_gen += 1                # a buffer moved: graphs captured on the old address are dead
            t = torch.empty(n, dtype=dtype, device=self.device)
            self._ws[name] = t
        return t[:n].view(*shape)

    # ------------------------------------------------------------------------------------------ host prep
    def prepare_text(self, input_ids, attention_mask, labels, post_ids, alphas=None, keeps=None, row_alphas=None) -> StepState:
        """Host side of the text-only branch (ps-slm.py:459-468): apply the CPS draws (drop mask, alpha) to the
        sentencepiece ids, build the merge plan, and upload all integer inputs in one staging copy.  ``row_alphas`` (one list
        per utterance, same lengths as ``post_ids``): per-ROW smoothing instead of one alpha per utterance -- the form CPS
        insertions need (an inserted blank row is an exact one-hot, ps-slm.py:396-398); excludes alphas / keeps."""
        geo = self.geo
        B = len(post_ids)
        kept = []
        for u, ids in enumerate(post_ids):
            ids = np.asarray(ids, dtype=np.int64)
            if alphas is not None and keeps is not None:
                ids = ids[np.asarray(keeps[u], dtype=bool)]
            kept.append(ids)
        lens = np.array([len(k) for k in kept], dtype=np.int64)
        kk = self.proj.k
        # the "linear" projector concatenates k frames per row and drops the batch tensor's trailing seq_len % k frames
        # (projector.py:41-45); every utterance then owns len // k projector rows (ps-slm.py:482)
        Lmax = (int(lens.max()) // kk) * kk
        if self.shape_buckets is not None and Lmax > 0:
            Lmax = rup(Lmax, self.shape_buckets[1] * kk)      # zero posterior rows up to the bucket (they are never merged)
        if Lmax == 0:
            raise ValueError("text branch: every utterance has an empty pseudo-posterior (no sentencepiece ids left after "
                             "cleaning / CPS drops); the projector needs at least one row")
        Fa = B * Lmax
        Fap = rup(Fa, 64 * kk)
        pid = np.full(Fap, -1, dtype=np.int32)
        pal = np.zeros(Fap, dtype=np.float32)
        for u, ids in enumerate(kept):
            n = min(len(ids), Lmax)
            pid[u * Lmax: u * Lmax + n] = ids[:n]
            if row_alphas is not None:
                pal[u * Lmax: u * Lmax + n] = np.asarray(row_alphas[u], dtype=np.float32)[:n]
            elif alphas is not None:
                pal[u * Lmax: u * Lmax + n] = float(alphas[u])
        st = self._finish_prepare(input_ids, attention_mask, labels, lens // kk, Lmax // kk)
        st.dev["post_ids"] = self._upload("post_ids", pid, flush=False)
        st.dev["post_alpha"] = self._upload("post_alpha", pal, flush=False)
        st.Ra, st.Rap, st.Fap = Fa // kk, Fap // kk, Fap
        if labels is not None and not self.freeze_projector:
            self._pad_rows(st, flush=False)              # (the projector backward's row index travels with the rest)
        self._flush_uploads()                            # ONE H2D copy for the step's integer inputs
        return st

    def prepare_audio(self, input_ids, attention_mask, labels, input_features, input_feature_length, do_psd=True, fp32=None) -> StepState:
        """Audio branch (ps-slm.py:430-454, :469-473, :482): encoder -> CTC posterior -> PSD -> projector; with ``raw_features``
        (train_config.ctc_posterior=false, ps-slm.py:515-523) PSD's decisions still come from the posterior but the rows it keeps /
        averages are the encoder's output states, and those feed the projector.  A projector that concatenates k frames per row
        (``linear`` / ``cov1d-linear`` with encoder_projector_ds_rate = k) drops the batch tensor's trailing Lmax % k frames and
        gives every utterance len // k rows (projector.py:41-45, ps-slm.py:482)."""
        from .encoder import psd_on_device
        B, T, _ = input_features.shape
        if fp32 is None:                                                           # generate() with use_fp16 = false: fp32 encoder too
            fp32 = self.arith == "fp32" and labels is None                         # (an eval-mode forward asks for it explicitly)
        if fp32:
            from .encoder import encoder_posterior_fp32
            post, Te, _ = encoder_posterior_fp32(self, input_features, input_feature_length)      # the fp32 posterior
        else:
            post, Te = self._encoder_output(input_features, input_feature_length)  # the CTC head's logits
        fl = np.asarray(input_feature_length.cpu() if isinstance(input_feature_length, torch.Tensor) else input_feature_length)
        fl_dev = self._upload("feat_lens", fl.astype(np.int32))
        kk = self.proj.k
        src = self._ws["enc_outf"][: post.shape[0] * self.geo.enc_dim].view(post.shape[0], self.geo.enc_dim) if self.raw_features else None
        try:
            rows, new_lens, Lmax = psd_on_device(self, post, B, T, Te, fl_dev, do_psd, k=kk, feats=src, logits=not fp32)
        finally:
            if self.device.type == "cuda":             # the encoder's output buffers are free for the next pass from here on
                self._psd_done = self._psd_done or torch.cuda.Event()
                self._psd_done.record()
        st = self._finish_prepare(input_ids, attention_mask, labels, np.minimum(new_lens, Lmax) // kk, Lmax // kk)
        st.Fap = rows.shape[0]
        st.Ra, st.Rap = B * Lmax // kk, st.Fap // kk
        st.path = "audio"
        if labels is not None and not self.freeze_projector:
            self._pad_rows(st, flush=False)
        self._flush_uploads()
        st.dev["post"] = rows
        st.dev["psd_lens"] = new_lens
        self._projector_from_posterior(st)
        return st

    def _encoder_output(self, input_features, input_feature_length):
        """(CTC logits bf16 [B * Te, Kp], Te) of this batch: the pass prefetch_encoder() started for it, or one run here."""
        from .encoder import encoder_posterior
        B, T, _ = input_features.shape
        ahead, self._enc_ahead = self._enc_ahead, None
        if ahead is not None:
            # also when the pending pass belongs to another batch: it shares the encoder's workspace with the pass run below
            torch.cuda.current_stream().wait_event(ahead["ready"])
            lens = np.asarray(input_feature_length.cpu() if isinstance(input_feature_length, torch.Tensor) else input_feature_length)
            if ahead["feats"] is input_features and ahead["version"] == input_features._version and np.array_equal(ahead["lens"], lens):
                return ahead["post"], ahead["Te"]
        post, Te, _ = encoder_posterior(self, input_features, input_feature_length, want_post=False)
        self._enc_shapes[(B, T)] = self._buf_gen
        return post, Te

    def prefetch_encoder(self, input_features, input_feature_length) -> bool:
        """Starts the frozen encoder pass (SenseVoice.py:548-579 + CTC head) of the NEXT batch on a side stream, so that it runs
        under the current batch's decoder step instead of in front of its own: the encoder is frozen (no gradient, weights never
        change), so its output for batch i + 1 does not depend on step i, and its ~420 short launches fill the ramps and tails of
        the decoder's GEMM grids (measured: 40.1 -> 36.0 ms per audio-SFT step of 16 utterances, `tools/lab_audio_overlap.py`).
        prepare_audio() of a batch whose ``input_features`` is this very tensor (unmodified, same lengths) waits for the pass and
        skips its own; any other batch runs the encoder as before (after waiting: the workspace is shared).  Same kernels on the
        same data: results are bit-identical with and without the call.  Returns False (and does nothing) on the CPU double,
        and -- with hipGraph replay on -- for a batch shape whose encoder graph does not exist yet (the first passes of a shape
        allocate and capture on the main stream)."""
        from .encoder import encoder_posterior
        if self.device.type != "cuda" or self.encoder is None or input_features is None:
            return False
        B, T, _ = input_features.shape
        if self.use_graphs and (self._enc_shapes.get((B, T)) != self._buf_gen or ("region", "encoder", B, T, False) not in self._graphs):
            return False                                  # graph replay: the first passes of a shape allocate and capture in line
        # (eager launches: any shape -- dynamic batching gives every batch its own.  A workspace buffer that has to grow is
        #  re-allocated inside the pass; every reader of the encoder's buffers is ordered against it by the two events)
        if self._enc_stream is None:
            from .streams import side_stream
            self._enc_stream = side_stream(self.device, "encoder one batch ahead", owner=self)   # a stream on another hardware queue than the current one
        side = self._enc_stream
        if self._psd_done is not None:
            side.wait_event(self._psd_done)               # PSD of the batch in flight has read the logits / encoder states
        lens = (input_feature_length.cpu().numpy() if isinstance(input_feature_length, torch.Tensor) else np.asarray(input_feature_length)).copy()
        with torch.cuda.stream(side):
            post, Te, _ = encoder_posterior(self, input_features, input_feature_length, want_post=False)
            ready = torch.cuda.Event()
            ready.record(side)
        self._enc_ahead = dict(feats=input_features, version=input_features._version, lens=lens, post=post, Te=Te, ready=ready)
        return True

    def _upload(self, name, arr, flush=True):
        """Host array -> its named device buffer (valid until the next upload under that name).  On the GPU through the pinned
        pack (UploadPack); ``flush=False`` defers the copy to the next ``_flush_uploads()`` (the prepare_* functions batch the
        step's 14 arrays into one copy)."""
        if self.device.type != "cuda" or os.environ.get("TASU_UPLOAD_PACK", "1") == "0":     # (=0: the per-array copies, for A/B runs)
            t = torch.from_numpy(np.ascontiguousarray(arr))
            d = self._buf("in_" + name, t.shape, t.dtype)
            d.copy_(t, non_blocking=True)
            return d
        if self._pack is None:
            self._pack = UploadPack(self)
        d = self._pack.put(name, arr)
        if flush:
            self._pack.flush()
        return d

    def _flush_uploads(self):
        if self._pack is not None:
            self._pack.flush()

    def _finish_prepare(self, input_ids, attention_mask, labels, num_audio, Lmax) -> StepState:
        to_np = lambda t: t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
        ids_np, am_np = to_np(input_ids), to_np(attention_mask).astype(bool)
        lab_np = None if labels is None else to_np(labels)
        n_mult = 64
        if self.shape_buckets is not None and lab_np is not None and ids_np.shape[0] > 1 and bool(am_np[:, 0].all()):
            # right-padded training batch: extra pad columns (pad id, mask 0, label -100) up to the next bucket of S
            s_mult, _, n_mult = self.shape_buckets
            S_now = ids_np.shape[1] - 1 + int(np.max(num_audio))
            extra = (-S_now) % s_mult
            if extra:
                B0 = ids_np.shape[0]
                ids_np = np.concatenate([ids_np, np.full((B0, extra), self.geo.eos_id, dtype=ids_np.dtype)], 1)
                am_np = np.concatenate([am_np, np.zeros((B0, extra), dtype=bool)], 1)
                lab_np = np.concatenate([lab_np, np.full((B0, extra), -100, dtype=lab_np.dtype)], 1)
        plan = build_merge_plan(ids_np, am_np, lab_np, num_audio, self.geo.speech_id, Lmax)
        st = StepState(plan=plan, B=plan.B, S=plan.S, M=plan.B * plan.S)
        st.dev["kind"] = self._upload("kind", plan.src_kind, flush=False)
        st.dev["idx"] = self._upload("idx", plan.src_idx, flush=False)
        st.dev["key_mask"] = self._upload("key_mask", plan.key_mask, flush=False)
        st.dev["pos"] = self._upload("pos", plan.position_ids, flush=False)
        st.dev["shift_labels"] = self._upload("shift_labels", plan.shift_labels, flush=False)
        st.dev["audio_rows"] = self._upload("audio_rows", plan.audio_rows, flush=False)
        st.dev["inv_count"] = self._upload("inv_count", np.array([1.0 / max(plan.count, 1)], dtype=np.float32), flush=False)
        if labels is not None:
            # compact index of the labelled positions: the training step's lm_head, CE and lm_head dgrad only run over these
            # rows (forward_llm, "labelled rows"); rows -> positions, slot = the inverse map, labels in compact order
            sl = np.asarray(plan.shift_labels).reshape(-1)
            rows = np.nonzero(sl >= 0)[0].astype(np.int32)
            st.nL, st.nLp = len(rows), rup(max(len(rows), 1), n_mult)
            lab_rows = np.full(st.nLp, -1, dtype=np.int32)
            lab_rows[: st.nL] = rows
            lab_c = np.full(st.nLp, -100, dtype=np.int32)
            lab_c[: st.nL] = sl[rows]
            slot = np.full(st.M, -1, dtype=np.int32)
            slot[rows] = np.arange(st.nL, dtype=np.int32)
            st.dev["lab_rows"] = self._upload("lab_rows", lab_rows, flush=False)
            st.dev["lab_rows0"] = self._upload("lab_rows0", np.maximum(lab_rows, 0), flush=False)   # padding slots read row 0 (their results are ignored)
            st.dev["lab_compact"] = self._upload("lab_compact", lab_c, flush=False)
            st.dev["lab_slot"] = self._upload("lab_slot", slot, flush=False)
        return st

    # ------------------------------------------------------------------------------------------ forward
    def forward_projector_text(self, st: StepState):
        """pseudo-posterior rows -> projector (projector.py:149-151 / :38-49)."""
        ops, pr = self.ops, self.proj
        post = self._buf("post", (st.Fap, pr.Kp), torch.float32)
        ops.posterior_build(st.dev["post_ids"], st.dev["post_alpha"], post, st.Fap, pr.K)
        st.dev["post"] = post
        self._projector_from_posterior(st)

    def _projector_from_posterior(self, st):
        """linear-silu: LayerNorm(25055) -> Linear -> SiLU -> Linear;  linear: [k frames concatenated] Linear -> ReLU -> Linear;
        cov1d-linear: Conv1d(k, stride k) -> ReLU -> Linear -> ReLU -> Linear."""
        ops, pr = self.ops, self.proj
        if pr.is_ca:
            return self._forward_cross_attention(st)
        Fap, Rap, K, Kp, Hb, Do = st.Fap, st.Rap, pr.K, pr.Kp, pr.Hb, pr.Do
        post = st.dev["post"]
        xn = self._buf("xn", (Fap, Kp), torch.bfloat16)
        mean = rstd = None
        if pr.has_norm:
            mean = self._buf("ln_mean", (Fap,), torch.float32)
            rstd = self._buf("ln_rstd", (Fap,), torch.float32)
            ops.layernorm_fwd(post, pr.view(pr.p, "norm.weight"), pr.view(pr.p, "norm.bias"), xn, mean, rstd, Fap, K,
                              self.geo.ln_eps)
        else:
            ops.cast_bf16(post, xn)                            # autocast: the Linear rounds its input to bf16
        xrows = xn.view(Rap, pr.k * Kp)                        # k consecutive frames = one projector row
        c0 = a0 = None
        if pr.has_conv:
            # Conv1d(kernel = stride = k) over time = one GEMM over the k concatenated frames; pad output channels stay zero
            # (zero weight rows, zero bias), then ReLU (projector.py:66-69)
            c0 = self._buf("c0", (Rap, Kp), torch.bfloat16)
            ops.gemm(xrows, pr.view(pr.pb, "conv1d.weight"), c0, Rap, Kp, pr.k * Kp, bias=pr.view(pr.pb, "conv1d.bias"))
            a0 = self._buf("a0", (Rap, Kp), torch.bfloat16)
            ops.relu_fwd(c0, a0)
            xrows = a0
        h1 = self._buf("h1", (Rap, Hb), torch.bfloat16)
        ops.gemm(xrows, pr.view(pr.pb, pr.n_w1), h1, Rap, Hb, pr.kin * Kp, bias=pr.view(pr.pb, pr.n_b1))
        a1 = self._buf("a1", (Rap, Hb), torch.bfloat16)
        (ops.silu_fwd if pr.has_norm else ops.relu_fwd)(h1, a1)
        y2 = self._buf("y2", (Rap, Do), torch.bfloat16)
        ops.gemm(a1, pr.view(pr.pb, pr.n_w2), y2, Rap, Do, Hb, bias=pr.view(pr.pb, pr.n_b2))
        st.dev.update(xn=xn, ln_mean=mean, ln_rstd=rstd, h1=h1, a1=a1, y2=y2, c0=c0, a0=a0)

    # ---- cross-attention projector (EncoderProjectorCTCCA, projector.py:104-126; ps-slm.py:475-480)
    def _ca_tables(self):
        """(E [V, D], E^T [D, Vpad]) in bf16: the LLM's input embedding table, which is keys AND values of the projector (detached:
        ps-slm.py:476-478).  With tied embeddings these are the lm_head copies the decoder already holds."""
        llm, geo = self.llm, self.geo
        if geo.tied:
            return llm.head, llm.head_t
        if getattr(llm, "_ca_e", None) is None:
            V, D, Vp = geo.llm_vocab, geo.llm_dim, rup(geo.llm_vocab, 64)
            e = llm.embed.to(torch.bfloat16).contiguous()
            et = torch.zeros(D, Vp, dtype=torch.bfloat16, device=self.device)
            et[:, :V].copy_(e.t())
            llm._ca_e, llm._ca_et = e, et
        return llm._ca_e, llm._ca_et

    def _forward_cross_attention(self, st):
        """Q = W_q(posterior rows); per head: P = softmax(Q_h E_h^T / sqrt(d)) over the V2 embedding rows, z_h = P E_h; the
        projector output is the heads side by side.  Kept for the backward: the bf16 scores (8 x [rows, V2]) and the rows' softmax
        statistics, from which the fp32 probabilities autograd would have saved are recomputed."""
        ops, pr, geo, d = self.ops, self.proj, self.geo, st.dev
        bf = torch.bfloat16
        Fap, Rap, Kp, D = st.Fap, st.Rap, pr.Kp, pr.Do
        V, Vp, H = geo.llm_vocab, rup(geo.llm_vocab, 64), geo.ca_heads
        dh = D // H
        if D % H or dh % 64:
            raise NotImplementedError(f"cross-attention projector: head width llm_dim / {H} = {D / H:g} must be a multiple of 64 "
                                      "(the K dimension of the score GEMMs)")
        denom = float(dh) ** 0.5                               # `/ (d ** 0.5)` of projector.py:120
        xn = self._buf("xn", (Fap, Kp), bf)
        ops.cast_bf16(d["post"], xn)                           # autocast: the Linear rounds its input to bf16
        q = self._buf("ca_q", (Rap, D), bf)
        ops.gemm(xn, pr.view(pr.pb, "W_q.weight"), q, Rap, D, Kp)
        E, ET = self._ca_tables()
        # kept for the backward: the bf16 SCORES of every head and the rows' softmax statistics -- autograd saves the softmax's
        # fp32 output, which the backward recomputes from them; the bf16 P is only the second einsum's operand (transient)
        S = self._buf("ca_s", (H, Rap, Vp), bf)
        stats = self._buf("ca_stats", (H, Rap, 2), torch.float32)
        P = self._buf("ca_p", (Rap, Vp), bf)
        y2 = self._buf("y2", (Rap, D), bf)
        for h in range(H):
            hs = slice(h * dh, (h + 1) * dh)
            ops.gemm(q[:, hs], E[:, hs], S[h], Rap, V, dh)     # scores of head h against every embedding row
            ops.scale_softmax_rows(S[h], P, Rap, V, denom, stats[h])
            ops.gemm(P, ET[hs], y2[:, hs], Rap, dh, Vp)
        d.update(xn=xn, ca_q=q, ca_s=S, ca_stats=stats, y2=y2)

    def _backward_cross_attention(self, st, on_ready):
        ops, pr, geo, d = self.ops, self.proj, self.geo, st.dev
        bf, f32 = torch.bfloat16, torch.float32
        Rap, Kp, D = st.Rap, pr.Kp, pr.Do
        V, Vp, H = geo.llm_vocab, rup(geo.llm_vocab, 64), geo.ca_heads
        dh, denom = D // H, float(D // H) ** 0.5
        audio_rows = d["audio_rows_pad"] if "audio_rows_pad" in d else self._pad_rows(st)
        dy2 = self._buf("dy2", (Rap, D), bf)
        ops.merge_bwd(d["dx"], audio_rows, dy2, Rap, D)
        E, ET = self._ca_tables()
        dp = self._buf("ca_p", (Rap, Vp), bf)                  # (the forward's transient P buffer)
        ds = self._buf("ca_ds", (Rap, Vp), bf)
        dq = self._buf("ca_dq", (Rap, D), bf)
        for h in range(H):
            hs = slice(h * dh, (h + 1) * dh)
            ops.gemm(dy2[:, hs], E[:, hs], dp, Rap, V, dh)     # dP = dz_h E_h^T
            ops.softmax_bwd_rows(d["ca_s"][h], d["ca_stats"][h], dp, ds, Rap, V, denom)
            ops.gemm(ds, ET[hs], dq[:, hs], Rap, dh, Vp)       # dQ_h = dS E_h
        dq_t = self._buf("ca_dq_t", (D, Rap), bf)
        xn_t = self._buf("xn_t", (Kp, Rap), bf)
        ops.transpose(dq, dq_t, Rap, D, Rap, D)
        ops.transpose(d["xn"].view(Rap, Kp), xn_t, Rap, Kp, Rap, Kp)
        ops.gemm(dq_t, xn_t, pr.view(pr.g, "W_q.weight"), D, Kp, Rap, mode=GEMM_F32)
        if on_ready is not None:
            on_ready(0, pr.numel if self.lora is None else self.lora.base)

    def forward_llm(self, st: StepState, compute_loss=True, need_backward=True, logits_rows="all"):
        ops, geo, llm = self.ops, self.geo, self.llm
        B, S, M = st.B, st.S, st.M
        D, I, H, G, V = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads, geo.llm_vocab
        Spad, Vp, LDQ = st.plan.Spad, rup(V, 64), (H + 2 * G) * HD
        scale = HD ** -0.5
        L = geo.llm_layers
        d = st.dev
        bf, f32 = torch.bfloat16, torch.float32
        xs = self._buf("xs", (2 * L + 1, M, D), f32)          # x_in[l] = xs[2l], x_mid[l] = xs[2l+1], final = xs[2L]
        ops.embed_merge(llm.embed, d["y2"], d["kind"], d["idx"], xs[0], M, D)
        cos = self._buf("cos", (M, HD // 2), f32)
        sin = self._buf("sin", (M, HD // 2), f32)
        ops.rope_table(d["pos"], cos, sin, HD, geo.rope_theta)
        rstd = self._buf("rstd", (2 * L + 1, M), f32)
        qkv = self._buf("qkv", (L, M, LDQ), bf)
        ao = self._buf("ao", (L, M, H * HD), bf)
        lse = self._buf("lse", (L, B * H * Spad), f32)
        gu = self._buf("gu", (L, M, 2 * I), bf)
        xn = self._buf("xn_llm", (M, D), bf)
        act = self._buf("act", (M, I), bf)
        lora = self._lora_run
        tail = bool(self.tail_rows and compute_loss and need_backward and not self.keep_logits and logits_rows != "none"
                    and "lab_rows" in d and st.nLp > 0 and lora is None)
        drop = lora is not None and lora._drop_on(self.training)
        if drop:
            ops.rng_advance(self.lora.rng)                     # new masks for this micro-step (a launch: graph replays advance too)
        st.lora_drop = drop
        fb = dict(xs=xs, rstd=rstd, qkv=qkv, ao=ao, lse=lse, gu=gu, xn=xn, act=act, cos=cos, sin=sin)
        for l, w in enumerate(llm.layers):
            if lora is not None:
                lora.layer_fwd(st, l, w, fb, drop)
                continue
            x_in, x_mid, x_out = xs[2 * l], xs[2 * l + 1], xs[2 * l + 2]
            ops.rmsnorm_fwd(x_in, w["ln1"], xn, rstd[2 * l], geo.rms_eps)
            ops.gemm_qkv_rope(xn, w["wqkv"], w["bqkv"], qkv[l], cos, sin, M, H, G, D)      # bias + RoPE in the GEMM's epilogue
            ops.attn_fwd(qkv[l], None, d["key_mask"], ao[l], lse[l], B, S, H, G, scale, True)
            ops.gemm(ao[l], w["wo"], x_mid, M, D, H * HD, resid=x_in, mode=GEMM_RESID)
            if tail and l == L - 1:
                # the last layer's MLP on the nLp labelled rows (compact operands; row gathers by the plan's index)
                n = st.nLp
                xn_t = self._buf("xn_tail", (n, D), bf)
                rstd_t = self._buf("rstd_tail", (n,), f32)
                ops.rmsnorm_fwd_rows(x_mid, d["lab_rows"], w["ln2"], xn_t, rstd_t, geo.rms_eps)
                gu_t = self._buf("gu_tail", (n, 2 * I), bf)
                act_t = self._buf("act_tail", (n, I), bf)
                ops.gemm_gate_up_swiglu(xn_t, w["wgu"], gu_t, act_t, n, I, D)
                xmid_t = self._buf("xmid_tail", (n, D), f32)
                ops.embed_rows(x_mid, d["lab_rows0"], xmid_t, n, D)
                xout_t = self._buf("xout_tail", (n, D), f32)
                ops.gemm(act_t, w["wd"], xout_t, n, D, I, resid=xmid_t, mode=GEMM_RESID)
                d.update(rstd_tail=rstd_t, gu_tail=gu_t, xout_tail=xout_t)
                continue
            ops.rmsnorm_fwd(x_mid, w["ln2"], xn, rstd[2 * l + 1], geo.rms_eps)
            ops.gemm_gate_up_swiglu(xn, w["wgu"], gu[l], act, M, I, D)          # gate|up projection + SwiGLU epilogue
            ops.gemm(act, w["wd"], x_out, M, D, I, resid=x_mid, mode=GEMM_RESID)
        d.update(xs=xs, cos=cos, sin=sin, rstd=rstd, qkv=qkv, ao=ao, lse=lse, gu=gu)
        if logits_rows == "none":                              # decode prefill: the caller projects the last rows only
            return
        if compute_loss and need_backward and not self.keep_logits:
            return self._loss_on_labelled_rows(st)
        ops.rmsnorm_fwd(xs[2 * L], llm.norm, xn, rstd[2 * L], geo.rms_eps)
        logits = self._buf("logits", (M, Vp), bf)
        ops.gemm(xn, llm.head, logits, M, V, D)
        d.update(logits=logits)
        if not compute_loss:
            return
        row_loss = self._buf("row_loss", (M,), f32)
        row_hit = self._buf("row_hit", (M,), torch.int32)
        # per-row argmax (the reference's `preds`, ps-slm.py:533) only matters at labelled rows for the accuracy; the kernel
        # skips the scan of ignored rows when no argmax buffer is passed, which the training step does
        row_arg = None if need_backward else self._buf("row_arg", (M,), torch.int32)
        if need_backward and not self.keep_logits:
            dlogits = logits                                   # overwrite in place (throughput mode)
        elif need_backward:
            dlogits = self._buf("dlogits", (M, Vp), bf)
        else:
            dlogits = None
        ops.ce_fwd_bwd(logits, d["shift_labels"], M, V, row_loss, row_hit, row_arg, dlogits, d["inv_count"])
        res = self._buf("loss_out", (4,), f32)
        ops.ce_reduce(row_loss, row_hit, d["shift_labels"], M, res)
        d.update(dlogits=dlogits, loss_out=res, row_arg=row_arg)

    def _loss_on_labelled_rows(self, st: StepState):
        """Throughput form of the loss head (training step, ``keep_logits=False``): the shifted CE ignores every position
        without a label (loss_utils.py:49-71, ignore_index -100), so final norm -> lm_head -> CE -> dlogits run over the
        labelled rows only, gathered into a compact [nLp, D] operand (half of the 4096 rows of the benchmark batch, whose
        prompt and audio positions carry no label).  Loss, accuracy and every gradient equal the full-materialisation path's
        (rows without a label have dlogits == 0 there); ``outputs.logits`` does not exist in this mode."""
        ops, geo, llm, d = self.ops, self.geo, self.llm, st.dev
        D, V, L = geo.llm_dim, geo.llm_vocab, geo.llm_layers
        Vp, n = rup(V, 64), st.nLp
        bf, f32 = torch.bfloat16, torch.float32
        xn_c = self._buf("xn_lab", (n, D), bf)
        rstd_c = self._buf("rstd_lab", (n,), f32)
        if "xout_tail" in d:                                 # the last layer already produced the labelled rows compact
            ops.rmsnorm_fwd(d["xout_tail"], llm.norm, xn_c, rstd_c, geo.rms_eps)
        else:
            ops.rmsnorm_fwd_rows(d["xs"][2 * L], d["lab_rows"], llm.norm, xn_c, rstd_c, geo.rms_eps)
        logits = self._buf("logits", (n, Vp), bf)
        ops.gemm(xn_c, llm.head, logits, n, V, D)
        row_loss = self._buf("row_loss", (n,), f32)
        row_hit = self._buf("row_hit", (n,), torch.int32)
        ops.ce_fwd_bwd(logits, d["lab_compact"], n, V, row_loss, row_hit, None, logits, d["inv_count"])   # dlogits in place
        res = self._buf("loss_out", (4,), f32)
        ops.ce_reduce(row_loss, row_hit, d["lab_compact"], n, res)
        d.update(dlogits=logits, loss_out=res, row_arg=None, rstd_lab=rstd_c, labelled_only=True)
        d.pop("logits", None)

    # ------------------------------------------------------------------------------------------ backward
    def backward(self, st: StepState, on_ready=None, w1_chunks=1):
        """dgrad-only through the frozen decoder, then wgrad of the projector into the flat grad buffer."""
        self.backward_llm(st)
        if not self.freeze_projector:
            self.backward_projector(st, on_ready, w1_chunks)

    def backward_llm(self, st: StepState, span=None):
        """lm_head dgrad, final norm, 28 decoder layers (dgrad only: the LLM is frozen).  Leaves d(loss)/d(inputs_embeds)
        in the fp32 workspace buffer ``dx``.  ``span`` = (hi, lo): only layers hi - 1 .. lo (the loss head with the span that
        starts at the last layer) -- the adapters' gradient exchange cuts the backward into such spans (run_backward)."""
        ops, geo, llm = self.ops, self.geo, self.llm
        B, S, M = st.B, st.S, st.M
        D, I, H, G, V = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads, geo.llm_vocab
        Spad, Vp, LDQ = st.plan.Spad, rup(V, 64), (H + 2 * G) * HD
        scale = HD ** -0.5
        L = geo.llm_layers
        d = st.dev
        bf, f32 = torch.bfloat16, torch.float32
        xs, rstd, cos, sin = d["xs"], d["rstd"], d["cos"], d["sin"]
        dx = self._buf("dx", (M, D), f32)
        dxb = self._buf("dxb", (M, D), bf)
        dn = self._buf("dn", (M, D), bf)                       # gradient wrt a normed activation
        dact = self._buf("dact", (M, I), bf)
        dgu = self._buf("dgu", (M, 2 * I), bf)
        dao = self._buf("dao", (M, H * HD), bf)
        delta = self._buf("delta", (B * H * Spad,), f32)
        dqkv = self._buf("dqkv", (M, LDQ), bf)
        dkp = self._buf("dkp", (M, H * HD), f32)
        dvp = self._buf("dvp", (M, H * HD), f32)
        l_hi, l_lo = span if span is not None else (L, 0)
        # lm_head dgrad (K = Vpad: dlogits pad columns are zero) and final norm
        if l_hi < L:
            pass                                               # a later span: the loss head ran with the first one
        elif d.get("labelled_only"):
            dn_c = self._buf("dn_lab", (st.nLp, D), bf)
            # [nLp, D] outputs in 128 x 192 tiles behind K = Vp: when they cover at most half of the 256 CUs the K range is
            # split so that every CU works (2048 labelled rows x 1536: 128 tiles x 2 ranges)
            tiles = ((st.nLp + 127) // 128) * ((D + 191) // 192)
            ksplit = max(1, min(256 // tiles, 8))
            while ksplit > 1 and Vp % (64 * ksplit):
                ksplit -= 1
            if ksplit > 1 and D % 4 == 0:
                ws = self._buf("dn_lab_slabs", (ksplit, st.nLp, D), f32)
                ops.gemm_splitk(d["dlogits"], llm.head_t, dn_c, st.nLp, D, Vp, ksplit, ws)
            else:
                ops.gemm(d["dlogits"], llm.head_t, dn_c, st.nLp, D, Vp)
            if "xout_tail" in d:
                # final norm and the last layer's MLP on the compact rows; the post-attention norm's backward scatters the result
                # (+ the rows' residual gradient) back to all M rows for that layer's attention backward
                n, w = st.nLp, llm.layers[L - 1]
                dx_t = self._buf("dx_tail", (n, D), f32)
                dxb_t = self._buf("dxb_tail", (n, D), bf)
                ops.rmsnorm_bwd(dn_c, d["xout_tail"], llm.norm, d["rstd_lab"], dx_t, dxb_t, False)
                dact_t = self._buf("dact_tail", (n, I), bf)
                dgu_t = self._buf("dgu_tail", (n, 2 * I), bf)
                ops.gemm_dswiglu(dxb_t, w["wd_t"], d["gu_tail"], dgu_t, dact_t, n, I, D)
                ops.gemm(dgu_t, w["wgu_t"], dn_c, n, D, 2 * I)
                ops.rmsnorm_bwd_rows_resid(dn_c, xs[2 * L - 1], w["ln2"], d["rstd_tail"], d["lab_slot"], dx_t, dx, dxb)
            else:
                ops.rmsnorm_bwd_rows(dn_c, xs[2 * L], llm.norm, d["rstd_lab"], d["lab_slot"], dx, dxb)
        else:
            ops.gemm(d["dlogits"], llm.head_t, dn, M, D, Vp)
            ops.rmsnorm_bwd(dn, xs[2 * L], llm.norm, rstd[2 * L], dx, dxb, False)
        lora = self._lora_run
        bb = dict(dx=dx, dxb=dxb, dn=dn, dact=dact, dgu=dgu, dao=dao, delta=delta, dqkv=dqkv, dkp=dkp, dvp=dvp)
        for l in range(l_hi - 1, l_lo - 1, -1):
            w = llm.layers[l]
            if lora is not None:
                lora.layer_bwd(st, l, w, bb, bool(getattr(st, "lora_drop", False)))
                continue
            x_in, x_mid = xs[2 * l], xs[2 * l + 1]
            if not (l == L - 1 and "xout_tail" in d):          # (the compact tail above has done the last layer's MLP)
                ops.gemm_dswiglu(dxb, w["wd_t"], d["gu"][l], dgu, dact, M, I, D)
                ops.gemm(dgu, w["wgu_t"], dn, M, D, 2 * I)
                ops.rmsnorm_bwd(dn, x_mid, w["ln2"], rstd[2 * l + 1], dx, dxb, True)
            ops.gemm(dxb, w["wo_t"], dao, M, H * HD, D)
            # delta = rowsum(dO . O), dQ, dK / dV and the rotary embedding's backward behind one entry point: the single-pass kernels
            # for Spad <= 256 (delta inside the kernel; dkp / dvp = one fp32 partial per query head), else the tiled kernels
            ops.attn_bwd_fused(d["qkv"][l], d["key_mask"], dao, d["ao"][l], d["lse"][l], delta, cos, sin, dqkv, dkp, dvp, B, S, H, G, scale, True)
            ops.gemm(dqkv, w["wqkv_t"], dn, M, D, LDQ)
            ops.rmsnorm_bwd(dn, x_in, w["ln1"], rstd[2 * l], dx, dxb, True)
        if lora is not None:
            lora.join()                                        # the adapters' weight-gradient chains (side stream) are back
        d["dx"] = dx

    def grad_ranges(self, w1_chunks=1):
        """The flat gradient bucket as the ranges backward_projector completes, in completion order:
        [b1 | W2 | b2] (one contiguous tail), then ``w1_chunks`` row blocks of W1, then the input stage's parameters --
        [norm.weight | norm.bias] (linear-silu) or [conv1d.weight | conv1d.bias] (cov1d-linear).  The ranges tile [0, numel)
        exactly."""
        pr = self.proj
        end = pr.numel if self.lora is None else self.lora.base            # the projector's own tensors end here
        # the adapters: one range per span of decoder layers, in the order the backward completes them (last layers first)
        head = [] if self.lora is None else [(self.lora.layer_range[hi - 1][0], self.lora.layer_range[lo][1]) for hi, lo in self.lora_spans()]
        if self.freeze_projector:
            return head                                                    # the projector's part of the bucket is not touched
        if pr.is_ca:
            return head + [(0, end)]
        o_w1, o_b1 = pr.offsets[pr.n_w1][0], pr.offsets[pr.n_b1][0]
        ld = pr.kin * pr.Kp
        rows = [pr.Hb * i // w1_chunks for i in range(w1_chunks + 1)]
        out = [(o_b1, end)] + [(o_w1 + r0 * ld, o_w1 + r1 * ld) for r0, r1 in zip(rows[:-1], rows[1:])]
        return head + out + ([(0, o_w1)] if o_w1 > 0 else [])

    def backward_projector(self, st: StepState, on_ready=None, w1_chunks=1):
        """Merge backward + projector backward (projector.py:149-151 / :38-49 reversed): wgrads land in the flat fp32 bucket.
        ``on_ready(lo, hi)`` (the engine's gradient exchange) is called as soon as the kernels that complete the bucket range
        [lo, hi) have been launched, in the order of ``grad_ranges(w1_chunks)``; with ``w1_chunks`` > 1 the W1 wgrad -- 94 % of
        the bucket -- runs as that many row-block GEMMs so that its all-reduce starts before the projector's input-side work
        (dxn, LayerNorm parameter gradients) has run."""
        ops, pr, d = self.ops, self.proj, st.dev
        if pr.is_ca:
            return self._backward_cross_attention(st, on_ready)
        bf, f32 = torch.bfloat16, torch.float32
        ranges = self.grad_ranges(w1_chunks)
        if self.lora is not None:
            ranges = ranges[len(self.lora_spans()):]   # (the adapters' ranges belong to backward_llm: run_backward reports them)
        # merge backward: gradient rows that hold audio -> projector output gradient
        Rap, K, Kp, Hb, Do = st.Rap, pr.K, pr.Kp, pr.Hb, pr.Do
        kKp = pr.kin * Kp                                     # input width of the first Linear
        audio_rows = d["audio_rows_pad"] if "audio_rows_pad" in d else self._pad_rows(st)
        dy2 = self._buf("dy2", (Rap, Do), bf)
        ops.merge_bwd(d["dx"], audio_rows, dy2, Rap, Do)
        # Linear2: db2, dW2 = dy2^T a1, da1 = dy2 W2
        ops.colsum(dy2, pr.view(pr.g, pr.n_b2), Rap, Do)
        dy2_t = self._buf("dy2_t", (Do, Rap), bf)
        a1_t = self._buf("a1_t", (Hb, Rap), bf)
        ops.transpose(dy2, dy2_t, Rap, Do, Rap, Do)
        ops.transpose(d["a1"], a1_t, Rap, Hb, Rap, Hb)
        ops.gemm(dy2_t, a1_t, pr.view(pr.g, pr.n_w2), Do, Hb, Rap, mode=GEMM_F32)
        da1 = self._buf("da1", (Rap, Hb), bf)
        ops.gemm(dy2, pr.w2b_t, da1, Rap, Hb, Do)
        dh1 = self._buf("dh1", (Rap, Hb), bf)
        (ops.silu_bwd if pr.has_norm else ops.relu_bwd)(da1, d["h1"], dh1)
        # Linear1: db1, dW1 = dh1^T xn, dxn = dh1 W1
        ops.colsum(dh1, pr.view(pr.g, pr.n_b1), Rap, Hb)
        if on_ready is not None:
            on_ready(*ranges[0])
        dh1_t = self._buf("dh1_t", (Hb, Rap), bf)
        xn_t = self._buf("xn_t", (kKp, Rap), bf)
        xrows = d["a0"] if pr.has_conv else d["xn"].view(Rap, kKp)
        ops.transpose(dh1, dh1_t, Rap, Hb, Rap, Hb)
        ops.transpose(xrows, xn_t, Rap, kKp, Rap, kKp)
        gw1 = pr.view(pr.g, pr.n_w1)
        rows = [Hb * i // w1_chunks for i in range(w1_chunks + 1)]
        for i, (r0, r1) in enumerate(zip(rows[:-1], rows[1:])):
            ops.gemm(dh1_t[r0:r1], xn_t, gw1[r0:r1], r1 - r0, kKp, Rap, mode=GEMM_F32)
            if on_ready is not None:
                on_ready(*ranges[1 + i])
        if pr.has_conv:
            # input stage of cov1d-linear: da0 = dh1 W1, through the ReLU, then db0 and dW0 = dc0^T [k frames concatenated]
            da0 = self._buf("dxn", (Rap, Kp), bf)
            ops.gemm(dh1, pr.w1b_t, da0, Rap, Kp, Hb)
            dc0 = self._buf("dc0", (Rap, Kp), bf)
            ops.relu_bwd(da0, d["c0"], dc0)
            ops.colsum(dc0, pr.view(pr.g, "conv1d.bias"), Rap, Kp)
            kcKp = pr.k * Kp
            dc0_t = self._buf("dc0_t", (Kp, Rap), bf)
            xcat_t = self._buf("xcat_t", (kcKp, Rap), bf)
            ops.transpose(dc0, dc0_t, Rap, Kp, Rap, Kp)
            ops.transpose(d["xn"].view(Rap, kcKp), xcat_t, Rap, kcKp, Rap, kcKp)
            ops.gemm(dc0_t, xcat_t, pr.view(pr.g, "conv1d.weight"), Kp, kcKp, Rap, mode=GEMM_F32)
            if on_ready is not None:
                on_ready(*ranges[-1])
            return
        if not pr.has_norm:
            return                                     # the posterior carries no parameters: nothing upstream of W1
        dxn = self._buf("dxn", (Rap, Kp), bf)
        ops.gemm(dh1, pr.w1b_t, dxn, Rap, Kp, Hb)
        ws = self._buf("ln_ws", (2 * LN_BWD_SPLIT * K,), f32)
        ops.layernorm_bwd_params(dxn, d["post"], d["ln_mean"], d["ln_rstd"], pr.view(pr.g, "norm.weight"),
                                 pr.view(pr.g, "norm.bias"), ws, Rap, K)
        if on_ready is not None:
            on_ready(*ranges[-1])

    def _pad_rows(self, st, flush=True):
        rows = np.full(st.Rap, -1, dtype=np.int32)
        rows[: st.Ra] = st.plan.audio_rows
        st.dev["audio_rows_pad"] = self._upload("audio_rows_pad", rows, flush=flush)
        return st.dev["audio_rows_pad"]

    # ------------------------------------------------------------------------------------------ hipGraph replay
    def _graphed(self, key, fn, st):
        self._flush_uploads()                           # (no-op unless a caller deferred an upload and forgot it)
        if not (self.use_graphs and self.device.type == "cuda"):
            return fn()
        g = self._graphs.get(key)
        if g is not None and g[2] != self._buf_gen:
            # a workspace buffer has grown since the capture (a larger batch shape came by): the graph holds freed
            # addresses.  Drop every graph of that generation and start over for this key.
            self._graphs = {k: v for k, v in self._graphs.items() if v[2] == self._buf_gen}
            self._graph_seen.pop(key, None)
            g = None
        if g is not None:
            self._graphs[key] = self._graphs.pop(key)     # most recently used
            g[0].replay()
            for k, v in g[1].items():                 # the views the captured code published into st.dev
                st.dev.setdefault(k, v)
            return
        seen = self._graph_seen.get(key, 0)
        self._graph_seen[key] = seen + 1
        if seen < 1:
            return fn()                              # eager warm-up: buffer allocation, lazy kernel attributes
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        before = set(st.dev)
        gen = self._buf_gen
        # The cyclic garbage collector must not run inside a capture: an unreachable CUDAGraph of an earlier model (a cycle freed
        # at a moment of the collector's choosing) would be destroyed while this stream is capturing -- hipGraphDestroy then fails
        # with "operation not permitted when stream is capturing" inside a destructor and takes the process down (seen in bench.py
        # between two legs).  torch.cuda.graph collects once before the capture begins; nothing may be collected until it ends.
        import gc
        gc_was = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):   # other threads (RCCL watchdog) may call into HIP
                try:
                    fn()
                except BaseException:
                    # an exception that unwinds out of a capture takes the process down in ~CUDAGraph: say what it was first
                    import traceback
                    traceback.print_exc()
                    raise
        finally:
            if gc_was:
                gc.enable()
        if gen != self._buf_gen:                      # a buffer grew DURING the capture: do not keep the graph
            return
        self._graphs[key] = (graph, {k: v for k, v in st.dev.items() if k not in before}, gen)
        while len(self._graphs) > self.graph_cache_size:          # LRU: dicts keep insertion order, replays re-insert
            old = next(iter(self._graphs))
            del self._graphs[old]
            self._graph_seen.pop(old, None)
        graph.replay()

    def graphed_region(self, key, fn):
        """Runs ``fn`` (a launch sequence that depends only on ``key`` and on the contents of persistent workspace
        buffers) eagerly or, with graphs enabled, as a captured hipGraph keyed by ``key`` (+ the workspace generation)."""
        class _NoState:
            dev = {}
        self._graphed(("region",) + tuple(key), fn, _NoState())

    def _shape_key(self, st, tag):
        return (tag, st.path, st.B, st.S, st.Ra, st.Rap, st.Fap, st.nLp, self.keep_logits, self.lora is not None and self.training)

    def _mark_dropout(self, st):
        """Whether THIS step's forward draws LoRA dropout masks -- recorded on the step state by host code that runs on every
        call.  forward_llm sets the same flag, but its Python body only runs at capture time: on a hipGraph replay a fresh
        StepState would keep the default False, and a backward captured then would regenerate no masks."""
        st.lora_drop = bool(self._lora_run is not None and self._lora_run._drop_on(self.training))

    def run_forward_text(self, st, compute_loss=True, need_backward=True):
        """forward_projector_text + forward_llm, graph-replayed when enabled."""
        def fn():
            self.forward_projector_text(st)
            self.forward_llm(st, compute_loss=compute_loss, need_backward=need_backward)
        self._mark_dropout(st)
        self._graphed(self._shape_key(st, ("fwd_text", compute_loss, need_backward)), fn, st)

    def run_forward_llm(self, st, compute_loss=True, need_backward=True):
        """forward_llm alone (audio branch: the projector has already run eagerly behind the host-side PSD plan)."""
        self._mark_dropout(st)
        self._graphed(self._shape_key(st, ("fwd_llm", compute_loss, need_backward)),
                      lambda: self.forward_llm(st, compute_loss=compute_loss, need_backward=need_backward), st)

    def run_backward(self, st, on_ready=None, w1_chunks=1):
        """Backward of the last forward, graph-replayed when enabled.  With a gradient-exchange hook (``on_ready``, N > 1) only
        the decoder part is replayed as a graph; the projector tail (~20 launches) is launched eagerly so that the hook can
        chain its collectives between the wgrad kernels."""
        if "audio_rows_pad" not in st.dev and not self.freeze_projector:
            self._pad_rows(st)                       # H2D upload stays outside the captured region
        if getattr(st, "fp32", False):               # train_config.use_fp16 = false: the fp32 step (ps_slm_amd/train_fp32.py), eager
            from .train_fp32 import backward_fp32
            return backward_fp32(self, st, on_ready)
        # use_peft: the adapters' weight-gradient chains fork onto a side stream 112 times per backward.  Captured into a hipGraph
        # those forks become branches that the runtime places on streams of ITS choosing, and whether they then run next to the
        # main branch depended on what the process had done before: 45 ms per step as the fifth leg of bench.py, 82 ms as the first
        # workload of a process (tools/lab_lora_graph_order.py), 46 ms with eager launches wherever it ran.  The adapted backward is
        # therefore launched eagerly (its ~900 launches hide under 30 ms of kernels); the forward, which forks nothing, stays a graph.
        run = getattr(self, "_lora_run", None)
        eager = self.lora is not None and run is not None and getattr(run, "side", None) is not None and not self.lora_bwd_graphs
        graphed = (lambda key, fn: fn()) if eager else (lambda key, fn: self._graphed(key, fn, st))
        if on_ready is None:
            graphed(self._shape_key(st, "bwd"), lambda: self.backward(st))
        else:
            if self.lora is None:
                self._graphed(self._shape_key(st, "bwd_llm"), lambda: self.backward_llm(st), st)
            else:
                # use_peft: the decoder's backward as one graph per span of layers; after each span its adapters' gradients --
                # a contiguous range of the bucket (the layers are laid out in completion order) -- go on the wire under the
                # remaining spans (513 MB per step travel in this recipe, against the projector's 218)
                for hi, lo in self.lora_spans():
                    graphed(self._shape_key(st, ("bwd_llm", hi, lo)), lambda hi=hi, lo=lo: self.backward_llm(st, (hi, lo)))
                    on_ready(self.lora.layer_range[hi - 1][0], self.lora.layer_range[lo][1])
            if not self.freeze_projector:
                self.backward_projector(st, on_ready, w1_chunks)

    # ------------------------------------------------------------------------------------------ results
    def logits_view(self, st):
        """[B, S, V] view of the bf16 logits buffer (valid until the next forward); None in the throughput mode of the
        training step (``keep_logits=False``), which only projects the labelled rows."""
        if "logits" not in st.dev:
            return None
        V = self.geo.llm_vocab
        return st.dev["logits"].view(st.B, st.S, -1)[:, :, :V]
