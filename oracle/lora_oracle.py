"""TEST INFRASTRUCTURE ONLY -- the LoRA recipe restated on the CPU (never imported by ps_slm_amd).

peft 0.6.0 (the reference's pinned dependency, absent from /root/reference and from this image) wraps every targeted
``nn.Linear`` of the HF decoder in ``peft.tuners.lora.Linear``, whose forward is, per its published source,

    result = F.linear(x, W, bias)
    x = x.to(lora_A.weight.dtype)
    result += lora_B(lora_A(lora_dropout(x))) * scaling            # scaling = lora_alpha / r

``HandLoraLinear`` is that formula on the reference's own HF modules (Multitask/model/ps-slm.py:114-117 is the call site that
installs it; config at Multitask/aispeech_asr_config.py:41-50).  PARITY: pinned to the formula, unpinned against peft itself.
Dropout: torch's Philox stream is not reproducible from outside, so the module takes the keep/scale tensor from ``mask_fn`` --
the numpy restatement below of the counter-based mask of ps_slm_amd/csrc/lora.hip -- and the parity tests compare results
for the SAME mask.
"""
import numpy as np
import torch
from torch import nn

TARGETS = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")


def lora_keep_mask(seed, step, sid, n, p):
    """bool [n], True = kept: upper 32 bits of splitmix64-finalize((seed ^ step * GOLD ^ sid << 44) + idx * ODD) >= p * 2^32."""
    M64 = np.uint64
    with np.errstate(over="ignore"):
        key = M64(seed & 0xFFFFFFFFFFFFFFFF) ^ (M64(step & 0xFFFFFFFFFFFFFFFF) * M64(0x9E3779B97F4A7C15)) ^ (M64(sid) << M64(44))
        z = key + np.arange(n, dtype=np.uint64) * M64(0xD1B54A32D192ED03)
        z = (z ^ (z >> M64(30))) * M64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> M64(27))) * M64(0x94D049BB133111EB)
        z = z ^ (z >> M64(31))
    thr = min(int(float(np.float32(p)) * 4294967296.0), 4294967295)
    return (z >> M64(32)) >= M64(thr)


def lora_keep_scale(rng, sid, shape, p):
    """fp32 tensor of ``shape``: 1 / (1 - p) where kept, 0 where dropped; rng = (seed, step)."""
    n = int(np.prod(shape))
    keep = lora_keep_mask(int(rng[0]), int(rng[1]), sid, n, p)
    inv = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return torch.from_numpy(np.where(keep, inv, np.float32(0)).astype(np.float32)).reshape(*shape)


class HandLoraLinear(nn.Module):
    def __init__(self, base: nn.Linear, A, B, scaling, p=0.0, mask_fn=None):
        super().__init__()
        self.base = base
        self.lora_A = nn.Parameter(A.clone().float())      # [r, in]
        self.lora_B = nn.Parameter(B.clone().float())      # [out, r]
        self.scaling, self.p, self.mask_fn = float(scaling), float(p), mask_fn

    def forward(self, x):
        result = self.base(x)
        xd = x.to(self.lora_A.dtype)
        if self.p > 0.0 and self.mask_fn is not None:
            xd = xd * self.mask_fn(xd.shape)
        return result + nn.functional.linear(nn.functional.linear(xd, self.lora_A), self.lora_B) * self.scaling


def sid_of(layer, target):
    return layer * 8 + TARGETS.index(target)


def apply_hand_lora(llm, lora_sd, targets, scaling, p=0.0, rng=None):
    """Replaces the targeted Linears of an HF Qwen2ForCausalLM in place.  lora_sd: reference-named adapter tensors
    (``llm.base_model.model.model.layers.N.{self_attn|mlp}.T.lora_{A|B}.default.weight``).  Returns {key: Parameter}."""
    params = {}
    for l, layer in enumerate(llm.model.layers):
        for parent_name in ("self_attn", "mlp"):
            parent = getattr(layer, parent_name)
            for t in targets:
                if not hasattr(parent, t):
                    continue
                ka = f"llm.base_model.model.model.layers.{l}.{parent_name}.{t}.lora_A.default.weight"
                kb = ka.replace("lora_A", "lora_B")
                mask_fn = None
                if p > 0.0:
                    mask_fn = (lambda shape, _s=sid_of(l, t): lora_keep_scale(rng, _s, shape, p))
                mod = HandLoraLinear(getattr(parent, t), lora_sd[ka], lora_sd[kb], scaling, p, mask_fn)
                setattr(parent, t, mod)
                params[ka], params[kb] = mod.lora_A, mod.lora_B
    return params
