"""Drop-in for the reference's model plugin ``model/ps-slm.py:model_factory`` (selected with
``++model_config.file=ps_slm_amd/ps_slm.py:model_factory``; contract: Multitask/model/ps-slm.py:130-181, loaded by
Multitask/utils/model_utils.py:9-33).  Same names, argument meaning and error behaviour; the arithmetic runs in
the HIP kernels of libtasu_hip.so through ps_slm_amd.model.TasuModel.

What callers get (Multitask/utils/deepspeed_utils.py:205-236, Multitask/inference_batch.py:139-151):
  model(**batch) -> (outputs, acc)   outputs.loss (0-dim fp32 device tensor with a grad_fn in training mode), outputs.logits [B,S,V] (bf16 view)
  model.generate(**batch) -> LongTensor [B, n_new]
  model.parameters() / named_parameters() / train() / eval() / state_dict() / load_state_dict(strict=False) with the
  reference's checkpoint keys ``encoder_projector.{norm,ffn.0,ffn.2}.{weight,bias}`` (Multitask/utils/checkpoint_handler.py:169-182)

Autograd glue (Multitask/finetune_deepspeed.py:127-149, Multitask/utils/deepspeed_utils.py:205-236): the model is not
an ``nn.Module`` (the frozen LLM and encoder weights live in kernel layouts and are not listed), but in training mode
``outputs.loss`` IS the result of a ``torch.autograd.Function`` over the trainable leaves ``parameters()`` lists (views of the flat
fp32 master buffer under the reference's names, the SAME objects on every call).  ``loss.backward()`` -- directly or through any
engine -- runs the hand-scheduled HIP backward (``TasuModel.run_backward``) once and hands each leaf its slice of the flat
gradient bucket, scaled by the incoming gradient; an optimizer built over ``model.parameters()`` then updates the masters in
place and the next ``forward`` / ``generate`` refreshes the bf16 working copies.  So the reference's loop body trains this plugin
with ``torch.optim.AdamW`` or DeepSpeed as it stands.  ``ps_slm_amd.engine.TasuEngine`` (the DeepSpeed engine's surface:
``backward(loss)`` / ``step()``) stays the FAST path: it ignores the graph, overlaps the gradient exchange with the backward and
runs the fused AdamW kernel.  A loss computed in eval mode has no saved activations behind it: it is an ``EngineLoss`` whose
``backward()`` raises and says so.
"""
import json
import logging
import os
import re
import sys
from types import SimpleNamespace

import numpy as np
import torch

# The reference loads a plugin with SourceFileLoader under the module name "<file name>", outside any package
# (Multitask/utils/dataset_utils.py:14-25), so this file cannot use package-relative imports: make the package importable
# by its absolute name from wherever the file lies, then import it like any other client would.
_PKG_PARENT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG_PARENT not in sys.path:
    sys.path.insert(0, _PKG_PARENT)

from ps_slm_amd.model import Geometry, TasuModel  # noqa: E402

logger = logging.getLogger(__name__)

DEFAULT_SPEECH_TOKEN = "<speech>"
DEFAULT_IGNORE_TOKEN = -100


# ------------------------------------------------------------------------------------------------ tokenizers
class SyntheticLLMTokenizer:
    """Stand-in used when ``llm_path`` holds no tokenizer files (the benchmark box has none): whitespace-separated
    integer strings <-> ids; any other word -> a deterministic ordinary id (so that cleaned training targets, which hold letters
    only -- speech_dataset_large.py:164 -- still have one token per word).  Exposes what the model and the entrypoints read
    (ps-slm.py:27,133-140,672-674)."""

    def __init__(self, geo: Geometry):
        self.eos_token_id = geo.eos_id
        self.pad_token_id = geo.eos_id
        self.bos_token_id = None
        self.default_speech_token = geo.speech_id
        self.default_ignore_token = DEFAULT_IGNORE_TOKEN
        self.vocab_size = geo.llm_vocab

    def encode(self, text):
        top = max(2, min(self.eos_token_id, self.vocab_size))          # ordinary ids: below the special tokens
        return [int(t) if t.lstrip("-").isdigit() else 1 + (sum((i + 1) * ord(c) for i, c in enumerate(t)) % (top - 1)) for t in text.split()]

    def batch_decode(self, ids, skip_special_tokens=True, **kw):
        out = []
        for row in ids.tolist() if isinstance(ids, torch.Tensor) else ids:
            toks = [t for t in row if not (skip_special_tokens and t in (self.eos_token_id, self.default_speech_token))]
            out.append(" ".join(map(str, toks)))
        return out


class SyntheticSentencePiece:
    """Stand-in for Multitask/model/tokenizer.py:SenseVoiceTokenizer when the BPE model file is absent:
    "12 7 3" -> [12, 7, 3]; any other word -> a deterministic id in [1, V)."""

    def __init__(self, vocab_size):
        self._v = vocab_size

    def encode(self, text):
        return [int(t) % self._v if t.isdigit() else 1 + sum(map(ord, t)) % (self._v - 1) for t in text.split()]

    @property
    def vocab_size(self):
        return self._v


def setup_tokenizer(train_config, model_config, geo, **kwargs):
    path = model_config.get("llm_path", None)
    if path and os.path.isfile(os.path.join(path, "tokenizer_config.json")):
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(path)
        tok.pad_token_id = tok.eos_token_id
        tok.add_special_tokens({"additional_special_tokens": [DEFAULT_SPEECH_TOKEN]})
        tok.default_ignore_token = DEFAULT_IGNORE_TOKEN
        tok.default_speech_token = tok.convert_tokens_to_ids(DEFAULT_SPEECH_TOKEN)
        return tok
    return SyntheticLLMTokenizer(geo)


def setup_encoder_tokenizer(model_config, geo):
    path = model_config.get("encoder_path", None)
    if path and os.path.isfile(os.path.join(path, "chn_jpn_yue_eng_ko_spectok.bpe.model")):
        import sentencepiece as spm
        sp = spm.SentencePieceProcessor(model_file=os.path.join(path, "chn_jpn_yue_eng_ko_spectok.bpe.model"))
        return SimpleNamespace(encode=lambda t: sp.encode(t, out_type=int), vocab_size=sp.vocab_size())
    return SyntheticSentencePiece(geo.ctc_vocab)


# ------------------------------------------------------------------------------------------------ geometry / weights
def geometry_from_config(model_config, raw_features=False) -> Geometry:
    """HF config.json under llm_path when present; otherwise a named synthetic geometry
    (``llm_path = synthetic:qwen2.5-1.5b | synthetic:qwen2.5-7b | synthetic:mid``)."""
    path = str(model_config.get("llm_path", "") or "")
    cfg_file = os.path.join(path, "config.json")
    if os.path.isfile(cfg_file):
        c = json.load(open(cfg_file))
        hd = c.get("head_dim", c["hidden_size"] // c["num_attention_heads"])
        if hd != 128:
            raise ValueError(f"head_dim {hd} is not supported by the gfx950 attention kernels (128 only)")
        geo = Geometry(llm_vocab=c["vocab_size"], llm_dim=c["hidden_size"], llm_inter=c["intermediate_size"],
                       llm_layers=c["num_hidden_layers"], llm_heads=c["num_attention_heads"],
                       llm_kv_heads=c["num_key_value_heads"], rope_theta=float(c.get("rope_theta", 1e6)),
                       rms_eps=float(c.get("rms_norm_eps", 1e-6)), tied=bool(c.get("tie_word_embeddings", True)))
    elif path.startswith("synthetic:"):
        name = path.split(":", 1)[1].lower()
        if name in ("qwen2.5-1.5b", "1.5b"):
            geo = Geometry.qwen25_1p5b()
        elif name in ("qwen2.5-7b", "7b"):
            geo = Geometry.qwen25_7b()
        elif name == "mid":
            from ps_slm_amd.synthetic import MID_GEOMETRY
            geo = Geometry.from_dict(MID_GEOMETRY)
        else:
            raise ValueError(f"unknown synthetic geometry {name!r}")
    else:
        raise FileNotFoundError(f"model_config.llm_path={path!r}: no config.json there and not a 'synthetic:<name>' spec")
    apply_encoder_config(geo, model_config.get("encoder_path", None))
    if raw_features:
        # train_config.ctc_posterior=false (ps-slm.py:515-523): model_config.encoder_dim is the width of the encoder states the
        # projector reads; the CTC vocabulary (PSD still decides on the posterior) stays the encoder's own
        ed = model_config.get("encoder_dim", None)
        if ed not in (None, geo.enc_dim) and path.lower() != "synthetic:mid":
            raise ValueError(f"ctc_posterior=false: model_config.encoder_dim={ed} must be the encoder's output size {geo.enc_dim}")
        geo.proj_in = geo.enc_dim
    elif model_config.get("encoder_dim", None) not in (None, geo.ctc_vocab) and path.lower() != "synthetic:mid":
        geo.ctc_vocab = int(model_config.encoder_dim)
    if model_config.get("llm_dim", None) not in (None, geo.llm_dim):
        raise ValueError(f"model_config.llm_dim={model_config.llm_dim} does not match the LLM hidden size {geo.llm_dim}")
    return geo


def apply_encoder_config(geo: Geometry, encoder_path):
    """funasr ships the encoder geometry in ``<encoder_path>/config.yaml`` (``input_size`` + ``encoder_conf``: the file
    funasr's AutoModel builds SenseVoiceSmall from, Multitask/model/ps-slm.py:91-106); absent file = the published
    SenseVoiceSmall values already in ``Geometry``."""
    cfg_file = os.path.join(str(encoder_path or ""), "config.yaml")
    if not os.path.isfile(cfg_file):
        return
    import yaml
    c = yaml.safe_load(open(cfg_file)) or {}
    ec = c.get("encoder_conf", {}) or {}
    for key, field in (("output_size", "enc_dim"), ("attention_heads", "enc_heads"), ("linear_units", "enc_ffn"),
                       ("num_blocks", "enc_blocks"), ("tp_blocks", "enc_tp_blocks"), ("kernel_size", "enc_kernel")):
        if key in ec:
            setattr(geo, field, int(ec[key]))
    if "input_size" in c:
        geo.feat_dim = int(c["input_size"])
    if int(ec.get("sanm_shfit", 0)) != 0:
        raise NotImplementedError("sanm_shfit != 0 is not what SenseVoiceSmall ships (SenseVoice.py:209-228 path only)")


def load_hf_llm_state_dict(path):
    """Reads HF Qwen2 ``*.safetensors`` shards under ``path`` into reference-named (``llm.``-prefixed) CPU tensors."""
    from safetensors.torch import load_file
    sd = {}
    for fn in sorted(os.listdir(path)):
        if fn.endswith(".safetensors"):
            for k, v in load_file(os.path.join(path, fn)).items():
                sd["llm." + k] = v.float()
    if not sd:
        raise FileNotFoundError(f"no .safetensors shards under {path}")
    return sd


def model_factory(train_config, model_config, **kwargs):
    """Same contract as Multitask/model/ps-slm.py:130-181: returns (model, tokenizer)."""
    projector = model_config.get("encoder_projector", "linear-silu")
    if projector == "q-former":
        # FINDING: the reference cannot run its q-former either.  EncoderProjectorQFormer.forward(x, atts) (projector.py:91-100)
        # takes the attention mask as a second argument and has no `.k`, but every call site of slam_model_asr passes one argument
        # and divides the lengths by `.k` (ps-slm.py:482,488,505,522, :612-651) -- TypeError "missing 1 required positional
        # argument: 'atts'"; with cross_attn=true the embedding table lands in `atts` and the merge fails on the 64 query rows
        # (reproduced with the imported reference: tests/test_oracle_golden.py::test_reference_cannot_run_its_q_former).
        raise NotImplementedError("encoder_projector='q-former': the reference's slam_model_asr cannot call EncoderProjectorQFormer "
                                  "(Multitask/model/projector.py:91 needs `atts`, ps-slm.py:482 passes one argument and reads `.k`); "
                                  "there is no behaviour to reproduce")
    if projector not in ("linear-silu", "linear", "cov1d-linear", "cross-attention"):
        raise NotImplementedError(f"encoder_projector={projector!r}: the MI355X path serves 'linear-silu' (the shipped recipe, "
                                  "Multitask/scripts/finetune_deespeed_sensevoice.sh:25), 'linear' (EncoderProjectorConcat, "
                                  "Multitask/model/projector.py:28-49), 'cov1d-linear' (EncoderProjectorCov1d, :53-73) and "
                                  "'cross-attention' (EncoderProjectorCTCCA, :104-126); q-former / simple_linear are not built")
    if not train_config.get("freeze_llm", True):
        raise NotImplementedError("the MI355X path keeps the decoder's own weights frozen (freeze_llm=true: Multitask/scripts/"
                                  "finetune_deespeed_sensevoice.sh:84); with use_peft=true the LoRA adapters train, full fine-tuning of "
                                  "the LLM is not built")
    if train_config.get("quantization", False) or train_config.get("use_emb", False):
        raise NotImplementedError("train_config.quantization / use_emb (ps-slm.py:101-102, :119-123) are not built")
    if kwargs.get("peft_ckpt", None):
        raise NotImplementedError("peft_ckpt (a peft adapter DIRECTORY, ps-slm.py:110-112): load the adapters from the training "
                                  "checkpoint with ckpt_path instead -- it holds them under the reference's own key names")
    fp32_mode = not train_config.get("use_fp16", False)
    if fp32_mode:
        # the reference computes in fp32 unless use_fp16 wraps the step in bf16 autocast (deepspeed_utils.py:160,205) and ALWAYS
        # decodes in fp32 (inference_batch.py:113-117).  use_fp16 = false therefore selects the fp32 kernels for generate()
        # (ps_slm_amd/decode_fp32.py), the eval-mode forward and the training step (ps_slm_amd/train_fp32.py); use_fp16 = true
        # selects bf16 autocast semantics, the path the benchmarks measure
        logger.warning("train_config.use_fp16 is false: the reference's fp32 arithmetic -- generate(), evaluation and the training step run on "
                       "the fp32 kernels (correctness mode: the training step is ~10x slower than with use_fp16=true, which selects the "
                       "bf16-autocast path the benchmarks measure); LoRA and the non-default projectors have the bf16 path only")
    raw = not train_config.get("ctc_posterior", True)
    if raw and projector == "cross-attention":
        raise NotImplementedError("ctc_posterior=false with the cross-attention projector: the reference's raw-feature branch "
                                  "(ps-slm.py:515-523) calls the projector with one argument, EncoderProjectorCTCCA needs two")
    geo = geometry_from_config(model_config, raw_features=raw)
    geo.projector = projector
    geo.projector_ds_rate = int(model_config.get("encoder_projector_ds_rate", 1) or 1) if projector in ("linear", "cov1d-linear") else 1
    tokenizer = setup_tokenizer(train_config, model_config, geo, **kwargs)
    if not isinstance(tokenizer, SyntheticLLMTokenizer):
        geo.speech_id, geo.eos_id = tokenizer.default_speech_token, tokenizer.eos_token_id
    device = kwargs.get("device", None)
    if device is None:
        device = f"cuda:{int(os.environ.get('LOCAL_RANK', train_config.get('device', 0) or 0))}"
    ops = kwargs.get("ops", None)
    if ops is None:
        from ps_slm_amd.ops import HipOps     # raises if libtasu_hip.so is missing or there is no GPU: no fallback
        ops = HipOps()
    core = TasuModel(geo, ops, device, keep_logits=bool(kwargs.get("keep_logits", True)))
    if fp32_mode and device != "cpu" and not str(device).startswith("cpu") and projector == "linear-silu" and not train_config.get("use_peft", False):
        core.arith = "fp32"
        core.llm.keep_f32 = True                       # fp32 copies of the frozen weights next to the bf16 ones (before loading)
    llm_path = str(model_config.get("llm_path", ""))
    need_encoder = not train_config.get("gt_emb", False) or raw or bool(kwargs.get("with_encoder", False))
    if llm_path.startswith("synthetic:"):
        core.init_random(seed=int(kwargs.get("init_seed", 1234)), with_encoder=need_encoder)
    else:
        sd = load_hf_llm_state_dict(llm_path)
        core.llm.load_reference_state_dict(sd)
        core.init_projector_default(seed=int(train_config.get("seed", 42)))
        enc_pt = os.path.join(str(model_config.get("encoder_path", "")), "model.pt")
        if need_encoder:
            core.load_encoder_checkpoint(enc_pt)
    # train_config.freeze_projector (the shipped script carries the knob, finetune_deespeed_sensevoice.sh:46,80): the reference
    # honours it for the linear-silu projector only (ps-slm.py:47-54) and then trains whatever else requires a gradient
    frozen_proj = bool(train_config.get("freeze_projector", False)) and projector == "linear-silu"
    if frozen_proj and not train_config.get("use_peft", False):
        raise ValueError("freeze_projector=true with use_peft=false leaves nothing to train (freeze_llm and freeze_encoder are fixed "
                         "here): the reference's optimizer would receive an empty parameter list")
    core.freeze_projector = frozen_proj
    if train_config.get("use_peft", False):
        from ps_slm_amd.lora import LoraConfig
        core.enable_lora(LoraConfig.from_peft_config(train_config.get("peft_config", {}) or {}), seed=int(train_config.get("seed", 42)) + 7)
        logger.info("LoRA: r=%d alpha=%g dropout=%g on %s: %d adapter parameters", core.lora.cfg.r, core.lora.cfg.lora_alpha,
                    core.lora.cfg.lora_dropout, ",".join(core.lora.cfg.target_modules), core.lora.num_parameters())
    model = slam_model_asr(core, tokenizer, setup_encoder_tokenizer(model_config, geo), train_config, model_config, **kwargs)
    ckpt_path = kwargs.get("ckpt_path", None)
    if ckpt_path is not None:
        logger.info("loading other parts from: %s", ckpt_path)
        model.load_state_dict(torch.load(ckpt_path, map_location="cpu"), strict=False)
    return model, tokenizer


# ------------------------------------------------------------------------------------------------ the model
NO_GRAPH_MSG = ("ps_slm_amd: this loss was computed in eval mode (model.eval() / need_backward=False): the step kept no activations "
                "for a backward pass.  Call model.train() before the forward whose loss you want to differentiate")


class EngineLoss(torch.Tensor):
    """``outputs.loss`` of an EVAL-mode forward: the 0-dim fp32 device tensor the step's CE kernel wrote.  It behaves like any
    tensor (``loss / k``, ``.detach().float()``, ``.item()`` of Multitask/utils/deepspeed_utils.py:283-293), results of arithmetic on
    it stay ``EngineLoss``, and ``backward()`` raises: no activations were saved.  (The TRAINING-mode loss is an ordinary tensor
    with a grad_fn: ``_HipStep`` below.)"""

    @staticmethod
    def __new__(cls, t):
        return torch.Tensor._make_subclass(cls, t.detach(), False)

    def backward(self, *args, **kwargs):
        raise RuntimeError(NO_GRAPH_MSG)


class _HipStep(torch.autograd.Function):
    """The autograd node behind ``outputs.loss``: forward = the loss the HIP step already computed; backward = the hand-scheduled HIP
    backward of that step (``TasuModel.run_backward``: dgrad through the frozen decoder, wgrad of the projector / the adapters into
    the flat fp32 bucket ``core.proj.g``), each leaf receiving ``grad_output x`` its slice of the bucket.  Nothing of the arithmetic
    runs in torch; the Function only shuttles pointers (SURVEY 8b "Autograd glue")."""

    @staticmethod
    def forward(ctx, loss_dev, model, st, *leaves):
        ctx.model, ctx.st = model, st
        return loss_dev.detach().clone()

    @staticmethod
    def backward(ctx, grad_out):
        model, st = ctx.model, ctx.st
        if model.last_state is not st:
            raise RuntimeError("ps_slm_amd: loss.backward() of an OLDER step -- the activations of a forward pass live in the model's "
                               "workspace until the next forward; differentiate each loss before running the next batch")
        if getattr(st, "backward_ran", False):
            raise RuntimeError("ps_slm_amd: this step's backward has already run (the workspace does not keep a graph for a second pass)")
        model.core.run_backward(st)
        st.backward_ran = True
        model._masters_touched = True                   # an optimizer is about to write the masters: refresh the bf16 copies next forward
        g = model.core.proj.g
        scale = grad_out.to(g.dtype)
        return (None, None, None) + tuple(v * scale for _, v in model._trainable_views(g))


class CausalLMOutput:
    __slots__ = ("loss", "logits")

    def __init__(self, loss, logits):
        self.loss, self.logits = loss, logits


class slam_model_asr:
    """Mirror of the reference's ``slam_model_asr`` (Multitask/model/ps-slm.py:183-537) over TasuModel."""

    def __init__(self, core: TasuModel, tokenizer, encoder_tokenizer, train_config, model_config, **kwargs):
        self.core = core
        self.tokenizer = tokenizer
        self.encoder_tokenizer = encoder_tokenizer
        self.train_config, self.model_config = train_config, model_config
        self.metric = kwargs.get("metric", "acc")
        self.ctc_posterior = train_config.get("ctc_posterior", True)
        self.do_psd = train_config.get("do_psd", True)
        self.voca_trans = train_config.get("voca_trans", False)
        self.gt_emb = train_config.get("gt_emb", False)
        self.gt_emb_noise = train_config.get("gt_emb_noise", False)
        if self.voca_trans or train_config.get("top1_emb", False):
            # FINDING: the reference's own voca_trans branch cannot run -- forward (ps-slm.py:485-514) and generate (:614-640) read
            # `encoder_outs`, which is only assigned in the voca_trans == False branch (:457-471, :590-603), and raise
            # UnboundLocalError (reproduced with the imported reference); top1_emb lives inside that branch.  There is nothing to
            # pin an implementation against.
            raise NotImplementedError("voca_trans / top1_emb: the reference's branch (Multitask/model/ps-slm.py:485-514) reads an "
                                      "unassigned `encoder_outs` and raises UnboundLocalError as shipped; not built")
        if not self.ctc_posterior:
            # the raw-feature branch (ps-slm.py:515-523) never looks at gt_emb: the encoder always runs
            self.gt_emb = False
        # knobs of ctc_pseudo_posterior_noise (ps-slm.py:372-375), overridable as attributes like in the reference
        self.drop_prob, self.insert_prob, self.smooth_low, self.smooth_high = 0.05, 0.0, 0.0, 0.1
        self.training = True
        self.last_state = None
        self._leaves = None                            # (key, [(name, leaf)]): named_parameters()
        self._masters_touched = False                  # a foreign optimizer may have written the masters since the last refresh

    def _refresh_if_touched(self):
        if self._masters_touched:
            self.core.sync_projector_copies()          # masters -> bf16 working copies + their transposes (TasuEngine.step() does this itself)
            self._masters_touched = False

    # ---- nn.Module-like surface
    def train(self, mode=True):
        self.training = mode
        self.core.training = bool(mode)                # LoRA dropout is active in training mode only
        return self

    def eval(self):
        return self.train(False)

    def to(self, *a, **k):
        return self

    def _views(self, flat):
        """(name, view of ``flat``, trainable) for every tensor ``named_parameters()`` lists: ``flat`` is the fp32 master buffer or the
        gradient bucket (same layout); K padding is sliced off where that is a plain column slice."""
        pr = self.core.proj
        for n in pr.names:
            v = pr.view(flat, n)
            r = pr.real[n]
            if v.dim() == 1:
                v = v[: r[0]]
            elif v.dim() == 2 and len(r) == 2 and not (n == pr.n_w1 and pr.kin > 1):
                v = v[:, : r[1]]
            yield "encoder_projector." + n, v, not self.core.freeze_projector
        if self.core.lora is not None:                 # use_peft=true: lora_A / lora_B of every adapted Linear, peft's key names
            for key, k in self.core.lora.names():
                yield key, self.core.lora.view(flat, *k), True

    def _trainable_views(self, flat):
        return [(n, v) for n, v, t in self._views(flat) if t]

    def named_parameters(self):
        """The trainable tensors under the reference's names (``encoder_projector.*``; what
        ``filter(lambda p: p.requires_grad, model.parameters())`` of Multitask/finetune_deepspeed.py:129 keeps): leaf views
        of the flat fp32 master buffer, ``requires_grad=True``, the SAME objects on every call (an optimizer built over them
        updates the masters in place; ``loss.backward()`` fills their ``.grad`` from the gradient bucket: ``_HipStep``)."""
        pr = self.core.proj
        key = (pr.p.data_ptr(), pr.p.numel(), self.core.lora is not None, bool(self.core.freeze_projector))
        if self._leaves is None or self._leaves[0] != key:
            flat = pr.p.detach()
            self._leaves = (key, [(n, v.requires_grad_(t)) for n, v, t in self._views(flat)])
        return iter(self._leaves[1])

    def parameters(self):
        return (p for _, p in self.named_parameters())

    def state_dict(self):
        """The trainable tensors (what the reference's checkpoint keeps: checkpoint_handler.py:169-182 saves with
        exclude_frozen_parameters): the projector and, with use_peft, the adapters."""
        proj = {} if self.core.freeze_projector else self.core.projector_state_dict()
        return {**proj, **self.core.lora_state_dict()}

    def load_state_dict(self, sd, strict=False):
        missing = []
        for n in self.core.proj.names:
            k = "encoder_projector." + n
            if k in sd:
                self.core.proj.load(n, sd[k].to(self.core.device, torch.float32))
            else:
                missing.append(k)
        known = set()
        if self.core.lora is not None:
            for key, k in self.core.lora.names():
                known.add(key)
                if key in sd:
                    self.core.lora.load(*k, sd[key])
                else:
                    missing.append(key)
        if strict and missing:
            raise KeyError(f"missing keys {missing}")
        self.core.sync_projector_copies()
        return missing, [k for k in sd if not k.startswith("encoder_projector.") and k not in known]

    # ---- CPS noise draws: same calls in the same order on the global CPU RNG as ps-slm.py:380-399
    def draw_noise(self, ids_list):
        """insert_prob == 0 (the shipped recipe): per utterance one uniform alpha, then one rand(len) keep mask."""
        if self.insert_prob != 0.0:
            raise ValueError("insert_prob != 0: use draw_noise_rows (insertions change the row sequence)")
        alphas, keeps = [], []
        for ids in ids_list:
            alphas.append(torch.empty(()).uniform_(self.smooth_low, self.smooth_high).item())
            keeps.append((torch.rand(len(ids)) > self.drop_prob).numpy())
        return alphas, keeps

    def draw_noise_rows(self, ids_list, blank_id=0):
        """The full CPS noise of ps-slm.py:360-409 as row descriptions: per utterance the final sequence of (id, alpha) after
        smoothing, drops and -- ``insert_prob`` > 0 -- int(len * insert_prob) insertions, each at randint(0, len + 1), each with
        probability 1/2 a copy of the row before it (same id, same alpha) and otherwise an exact one-hot blank row (alpha 0).
        Same calls in the same order on the global CPU RNG as the reference.  Returns (ids per utterance, alphas per row)."""
        out_ids, out_alpha = [], []
        for ids in ids_list:
            alpha = torch.empty(()).uniform_(self.smooth_low, self.smooth_high).item()
            keep = (torch.rand(len(ids)) > self.drop_prob).numpy()
            rows = [(int(i), alpha) for i, k in zip(ids, keep) if k]
            for _ in range(int(len(rows) * self.insert_prob)):
                pos = torch.randint(0, len(rows) + 1, (1,)).item()
                if bool(torch.rand(1) < 0.5) and len(rows) > 0:
                    rows.insert(pos, rows[pos - 1] if pos > 0 else rows[0])
                else:
                    rows.insert(pos, (int(blank_id), 0.0))
            out_ids.append([r[0] for r in rows])
            out_alpha.append([r[1] for r in rows])
        return out_ids, out_alpha

    # ---- forward (ps-slm.py:411-537)
    def forward(self, input_ids=None, input_features=None, attention_mask=None, input_feature_length=None, GT=None,
                labels=None, **unused):
        core = self.core
        self._refresh_if_touched()
        # use_fp16 = false outside training: the reference's fp32 arithmetic (evaluation(), deepspeed_utils.py:394-498, or any
        # model(**batch) under no autocast) -- fp32 projector / encoder / decoder / logits / CE (ps_slm_amd/decode_fp32.py)
        fp32_eval = core.arith == "fp32" and not self.training
        fp32_train = core.arith == "fp32" and self.training and labels is not None      # the shipped recipe trains in fp32 too (train_fp32.py)
        if self.gt_emb:
            ids_list = [self.encoder_tokenizer.encode(t) for t in GT]
            alphas = keeps = row_alphas = None
            if self.gt_emb_noise and self.insert_prob != 0.0:
                ids_list, row_alphas = self.draw_noise_rows(ids_list, core.geo.blank_id)
            elif self.gt_emb_noise:
                alphas, keeps = self.draw_noise(ids_list)
            st = core.prepare_text(input_ids, attention_mask, labels, ids_list, alphas, keeps, row_alphas=row_alphas)
            if not (fp32_eval or fp32_train):
                core.run_forward_text(st, compute_loss=labels is not None, need_backward=self.training)
        else:
            if input_features is None:
                raise ValueError("the audio branch needs input_features: dataset_config.text_only=true is only valid with "
                                 "train_config.gt_emb=true (text pseudo-posterior instead of the encoder)")
            st = core.prepare_audio(input_ids, attention_mask, labels, input_features, input_feature_length,
                                    do_psd=self.do_psd, fp32=fp32_eval or fp32_train)
            if not (fp32_eval or fp32_train):
                core.run_forward_llm(st, compute_loss=labels is not None, need_backward=self.training)
        if fp32_train:
            from ps_slm_amd.train_fp32 import forward_train_fp32
            forward_train_fp32(core, st)
        elif fp32_eval:
            from ps_slm_amd.decode_fp32 import forward_fp32
            forward_fp32(core, st, compute_loss=labels is not None)
        self.last_state = st
        if labels is None:
            return CausalLMOutput(None, core.logits_view(st)), -1
        res = st.dev["loss_out"]
        if self.training:
            leaves = [p for _, p in self.named_parameters() if p.requires_grad]
            loss = _HipStep.apply(res[0], self, st, *leaves)
        else:
            loss = EngineLoss(res[0])
        return CausalLMOutput(loss, core.logits_view(st)), (res[1] if self.metric else -1)

    __call__ = forward

    def prefetch(self, input_features=None, input_feature_length=None, **unused) -> bool:
        """Optional, audio branch only: start the frozen encoder pass of the NEXT batch now, on a side stream under the current
        batch's decoder step (TasuModel.prefetch_encoder).  Takes the collator's batch dict like forward(); the forward() of that
        very batch then skips its own encoder pass.  Results do not change; a batch that was not announced (or the text branch,
        or the CPU double) runs as before and the call returns False."""
        if self.gt_emb or input_features is None:
            return False
        return self.core.prefetch_encoder(input_features, input_feature_length)

    @torch.no_grad()
    def generate(self, input_ids=None, input_features=None, attention_mask=None, input_feature_length=None,
                 targets=None, **kwargs):
        from ps_slm_amd.decode import beam_search_generate
        core = self.core
        self._refresh_if_touched()
        # the decode loop is HF beam search with do_sample=False (ps-slm.py:660-675 defaults): a sampling / penalty knob set to
        # anything else would silently be ignored, so it is rejected
        for name, default in (("do_sample", False), ("top_p", 1.0), ("repetition_penalty", 1.0), ("temperature", 1.0)):
            if kwargs.get(name, default) != default:
                raise NotImplementedError(f"generate({name}={kwargs[name]!r}): the MI355X decode loop implements the reference's "
                                          f"defaults only ({name}={default!r}, Multitask/model/ps-slm.py:660-675: beam search "
                                          "without sampling, penalties or temperature)")
        if self.gt_emb:                                     # ps-slm.py:590-598
            texts = [re.sub(r"[^A-Za-z\s.,!?]+", "", t).lower().strip() for t in targets]
            ids_list = [self.encoder_tokenizer.encode(t) for t in texts]
            st = core.prepare_text(input_ids, attention_mask, None, ids_list, None, None)
            if core.arith != "fp32":                       # (the fp32 path runs its own fp32 projector)
                core.forward_projector_text(st)
        else:
            st = core.prepare_audio(input_ids, attention_mask, None, input_features, input_feature_length,
                                    do_psd=self.do_psd)
        if core.arith == "fp32":                           # use_fp16 = false: the reference's own decode arithmetic
            from ps_slm_amd.decode_fp32 import beam_search_generate_fp32 as beam_search_generate
        return beam_search_generate(core, st, num_beams=kwargs.get("num_beams", 4),
                                    max_new_tokens=kwargs.get("max_new_tokens", 200),
                                    min_length=kwargs.get("min_length", 1),
                                    length_penalty=kwargs.get("length_penalty", 1.0),
                                    eos_token_id=self.tokenizer.eos_token_id, pad_token_id=self.tokenizer.pad_token_id)
