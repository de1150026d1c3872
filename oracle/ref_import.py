"""TEST INFRASTRUCTURE ONLY -- imports the *actual* reference (``/root/reference``) in the build
container so that golden vectors can be generated from it (oracle/make_golden.py).

The reference depends on funasr / peft / omegaconf, none of which is installed here, so tiny stand-in
modules are injected into ``sys.modules`` first.  These stand-ins are OUR code (nothing is copied from the
reference): they only provide the *names* imported at Multitask/model/SenseVoice.py:10-17,
Multitask/model/ps-slm.py:16 and Multitask/utils/config_utils.py:9-15.  The only stand-in that carries
arithmetic is ``CTC`` (funasr's CTC head is an ``nn.Linear(encoder_output_size, odim)`` named ``ctc_lo``,
used at Multitask/model/ps-slm.py:450).

Never imported by the product path and never run on the GPU box (the reference tree does not exist there).
"""
import importlib.machinery
import importlib.util
import os
import sys
import types

import torch
from torch import nn

REFERENCE_ROOT = os.environ.get("TASU_REFERENCE_ROOT", "/root/reference")


class _Registry:
    """funasr.register.tables stand-in: ``@tables.register(table, key)`` + ``tables.<table>.get(key)``."""

    def __init__(self):
        for t in ("encoder_classes", "model_classes", "specaug_classes", "normalize_classes"):
            setattr(self, t, {})

    def register(self, table, key):
        def deco(cls):
            getattr(self, table)[key] = cls
            return cls

        return deco


class _CTC(nn.Module):
    def __init__(self, odim, encoder_output_size, **kw):
        super().__init__()
        self.ctc_lo = nn.Linear(encoder_output_size, odim)


class _AcceptAnything(nn.Module):
    def __init__(self, *a, **kw):
        super().__init__()


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    if "funasr" in sys.modules and getattr(sys.modules["funasr"], "_tasu_stub", False):
        return
    _mod("funasr", _tasu_stub=True)
    _mod("funasr.register", tables=_Registry())
    for pkg in ("funasr.models", "funasr.models.ctc", "funasr.models.paraformer", "funasr.utils",
                "funasr.train_utils", "funasr.losses", "funasr.metrics"):
        _mod(pkg)
    _mod("funasr.models.ctc.ctc", CTC=_CTC)
    _mod("funasr.utils.datadir_writer", DatadirWriter=object)
    _mod("funasr.models.paraformer.search", Hypothesis=object)
    _mod("funasr.train_utils.device_funcs", force_gatherable=lambda *a, **k: a)
    _mod("funasr.losses.label_smoothing_loss", LabelSmoothingLoss=_AcceptAnything)
    _mod("funasr.metrics.compute_acc", compute_accuracy=None, th_accuracy=None)
    _mod("funasr.utils.load_utils", load_audio_text_image_video=None, extract_fbank=None)
    _mod("peft", PeftModel=object, LoraConfig=object, TaskType=object, get_peft_model=None,
         prepare_model_for_kbit_training=None, AdaptionPromptConfig=object, PrefixTuningConfig=object)
    _mod("omegaconf", OmegaConf=object, DictConfig=dict, ListConfig=list)


def load_reference():
    """Returns (ps_slm module, SenseVoice module, projector module) of the real reference."""
    install_stubs()
    mt = os.path.join(REFERENCE_ROOT, "Multitask")
    if not os.path.isdir(mt):
        raise FileNotFoundError(f"reference tree not found at {mt}")
    if mt not in sys.path:
        sys.path.insert(0, mt)
    import model.SenseVoice as sv  # noqa: E402  (namespace package inside the reference)
    import model.projector as proj  # noqa: E402
    loader = importlib.machinery.SourceFileLoader("ref_ps_slm", os.path.join(mt, "model", "ps-slm.py"))
    spec = importlib.util.spec_from_loader("ref_ps_slm", loader)
    ps = importlib.util.module_from_spec(spec)
    loader.exec_module(ps)
    return ps, sv, proj


class Cfg(dict):
    """Config object supporting both attribute access and .get (the reference uses both)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class FakeSentencePiece:
    """Stand-in for SenseVoiceTokenizer (no BPE model file exists here): "text" is a string of
    space-separated integer ids, which is what the golden generator feeds as GT / targets."""

    def __init__(self, vocab_size):
        self._v = vocab_size

    def encode(self, text):
        # "12 7 3" -> [12, 7, 3]; any other word -> a deterministic id in [1, V)
        return [int(t) if t.isdigit() else 1 + sum(map(ord, t)) % (self._v - 1) for t in text.split()]

    @property
    def vocab_size(self):
        return self._v


class FakeLLMTokenizer:
    def __init__(self, speech_id, eos_id):
        self.default_speech_token = speech_id
        self.default_ignore_token = -100
        self.pad_token_id = eos_id
        self.eos_token_id = eos_id
        self.bos_token_id = None


def build_reference_model(geo, seed, train_flags, projector="linear-silu", ds_rate=1, projector_in=None):
    """Seeded random-init reference ``slam_model_asr`` at geometry ``geo`` (dict); ``projector``: "linear-silu"
    (EncoderProjectorLinearSiLU), "linear" (EncoderProjectorConcat with encoder_projector_ds_rate = ds_rate) or "cov1d-linear"
    (EncoderProjectorCov1d, kernel = stride = ds_rate) or "cross-attention" (EncoderProjectorCTCCA)."""
    import transformers

    ps, sv, proj = load_reference()
    torch.manual_seed(seed)
    enc = sv.SenseVoiceSmall(
        encoder="SenseVoiceEncoderSmall",
        encoder_conf=dict(output_size=geo["enc_dim"], attention_heads=geo["enc_heads"],
                          linear_units=geo["enc_ffn"], num_blocks=geo["enc_blocks"],
                          tp_blocks=geo["enc_tp_blocks"], kernel_size=geo["enc_kernel"], sanm_shfit=0,
                          dropout_rate=0.1),
        input_size=geo["feat_dim"], vocab_size=geo["ctc_vocab"])
    for p in enc.parameters():
        p.requires_grad = False
    enc.eval()
    qcfg = transformers.Qwen2Config(
        vocab_size=geo["llm_vocab"], hidden_size=geo["llm_dim"], intermediate_size=geo["llm_inter"],
        num_hidden_layers=geo["llm_layers"], num_attention_heads=geo["llm_heads"],
        num_key_value_heads=geo["llm_kv_heads"], max_position_embeddings=4096,
        rope_theta=geo.get("rope_theta", 1e6), rms_norm_eps=1e-6, tie_word_embeddings=geo.get("tied", True),
        attention_dropout=0.0, use_sliding_window=False)
    llm = transformers.Qwen2ForCausalLM(qcfg)
    for p in llm.parameters():
        p.requires_grad = False
    llm.eval()
    model_config = Cfg(encoder_projector=projector, encoder_path="/nonexistent", encoder_projector_ds_rate=ds_rate,
                       encoder_dim=projector_in or geo["ctc_vocab"], llm_dim=geo["llm_dim"])   # projector_in: the raw-feature branch reads encoder states
    if projector == "cross-attention":
        projector = proj.EncoderProjectorCTCCA(model_config)       # one matrix W_q; 8 heads over the LLM's embedding table
    elif projector == "cov1d-linear":
        projector = proj.EncoderProjectorCov1d(model_config)       # hidden width fixed at 2048 by the reference class
    elif projector == "linear":
        projector = proj.EncoderProjectorConcat(model_config)      # bottleneck fixed at 2048 by the reference class
    else:
        projector = proj.EncoderProjectorLinearSiLU(model_config, bottleneck=geo["bottleneck"])
    train_config = Cfg(ctc_posterior=True, do_psd=True, voca_trans=False, gt_emb=True, gt_emb_noise=False,
                       top1_emb=False)
    train_config.update(train_flags)
    tok = FakeLLMTokenizer(geo["speech_id"], geo["eos_id"])
    import model.tokenizer as reftok

    reftok.SenseVoiceTokenizer = lambda path: FakeSentencePiece(geo["ctc_vocab"])
    model = ps.slam_model_asr(enc, llm, projector, tok, train_config, model_config)
    return model
