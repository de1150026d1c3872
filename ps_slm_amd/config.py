"""Configuration surface of the two entrypoints: the same dataclass field names and defaults as the reference's
structured config (Multitask/aispeech_asr_config.py:26-181 + RunConfig in Multitask/finetune_deepspeed.py:19-37 /
Multitask/inference_batch.py:20-41) and the same ``++section.key=value`` command-line override syntax.  hydra /
omegaconf are not installed on the target image, so the ~100-line parser below replaces them.
Only the fields the hot path reads are interpreted; the rest are carried for compatibility.
"""
import dataclasses
import json
from dataclasses import dataclass, field
from typing import List, Optional


class _Section:
    """attribute access + ``.get`` (the reference uses both: ps-slm.py:207-223)."""

    def get(self, key, default=None):
        return getattr(self, key, default)

    def __getitem__(self, key):
        return getattr(self, key)

    def __contains__(self, key):
        return hasattr(self, key)

    def keys(self):
        return [f.name for f in dataclasses.fields(self)]


@dataclass
class ModelConfig(_Section):
    file: str = "ps_slm_amd/ps_slm.py:model_factory"
    llm_name: str = "Qwen2.5-7B-Instruct"
    llm_path: str = "PATH/to/LLAMA/7B"
    llm_type: str = "decoder_only"
    llm_dim: int = 4096
    encoder_name: str = "whisper"
    encoder_path: Optional[str] = None
    encoder_dim: int = 768
    encoder_projector: str = "linear"
    encoder_projector_ds_rate: int = 2
    ctc_linear: Optional[str] = None


@dataclass
class PeftConfig(_Section):
    """Multitask/aispeech_asr_config.py:41-50 (read by ps_slm_amd.lora.LoraConfig.from_peft_config when use_peft=true)."""
    peft_method: str = "lora"
    r: int = 64
    lora_alpha: int = 16
    target_modules: List[str] = field(default_factory=lambda: ["q_proj", "k_proj", "v_proj", "o_proj", "up_proj", "gate_proj", "down_proj"])
    bias: str = "none"
    task_type: str = "CAUSAL_LM"
    lora_dropout: float = 0.05
    inference_mode: bool = False


@dataclass
class TrainConfig(_Section):
    model_name: str = "asr_model"
    enable_ddp: bool = False
    enable_deepspeed: bool = False
    enable_fsdp: bool = False
    run_validation: bool = True
    batch_size_training: Optional[int] = None
    batching_strategy: str = "packing"
    context_length: int = 4096
    gradient_accumulation_steps: int = 1
    num_epochs: int = 3
    num_workers_dataloader: int = 1
    warmup_steps: int = 1000
    total_steps: int = 100000
    validation_interval: int = 1000
    lr: float = 1e-4
    weight_decay: float = 0.0
    seed: int = 42
    use_fp16: bool = False
    mixed_precision: bool = True
    val_batch_size: Optional[int] = None
    do_psd: bool = False
    ctc_posterior: Optional[bool] = False
    voca_trans: Optional[bool] = False
    use_peft: bool = False
    peft_config: PeftConfig = field(default_factory=PeftConfig)
    use_emb: bool = False
    gt_emb: bool = False
    gt_emb_noise: bool = False
    top1_emb: bool = False
    cross_attn: bool = False
    output_dir: str = "PATH/to/save/PEFT/model"
    freeze_projector: bool = False
    quantization: bool = False
    save_model: bool = True
    freeze_llm: bool = False
    freeze_encoder: bool = False
    device: Optional[int] = 0
    gaussian_sim: bool = False


@dataclass
class DataConfig(_Section):
    file: Optional[str] = "dataset/speech_dataset_large.py:get_speech_dataset"
    dataset: str = "multitask_dataset"
    encoder: str = "whisper"
    encoder_path: Optional[str] = None
    max_audio_length: int = 30
    train_max_frame_length: int = 1500
    ds_rate: int = 8
    eval_max_frame_length: int = 2000
    multitask_prompt_path: str = "conf/multiprompt.jsonl"
    prompt_style: str = "<|im_start|>user\n{}<speech><|im_end|>\n<|im_start|>assistant\n"
    append_info_tasks: List = field(default_factory=lambda: ["hotword"])
    train_scp_file_path: str = ""
    dev_scp_file_path: str = ""
    test_scp_file_path: str = ""
    inference_mode: bool = False
    text_only: bool = False          # ps_slm_amd/dataset.py: read audio lengths only (text-only alignment recipe)
    decode_threads: int = 4          # ps_slm_amd/dataset.py: threads that read + decode the next utterances' audio (0 / 1: in line)


@dataclass
class LogConfig(_Section):
    use_wandb: bool = False
    wandb_dir: str = "tmp/test_wandb"
    wandb_entity_name: str = "project_name"
    wandb_project_name: str = "project_name"
    wandb_exp_name: str = "exp_name"
    log_file: str = "tmp/test.log"
    log_interval: int = 5


@dataclass
class RunConfig(_Section):
    dataset_config: DataConfig = field(default_factory=DataConfig)
    model_config: ModelConfig = field(default_factory=ModelConfig)
    train_config: TrainConfig = field(default_factory=TrainConfig)
    log_config: LogConfig = field(default_factory=LogConfig)
    debug: bool = False
    metric: str = "acc"
    ckpt_path: Optional[str] = None
    deepspeed_config: str = ""
    decode_log: str = "output/decode_log"
    peft_ckpt: Optional[str] = None


def _coerce(text, current):
    """Parse an override value the way OmegaConf would for the target field's current type."""
    low = text.lower()
    if low in ("null", "none"):
        return None
    if isinstance(current, bool) or low in ("true", "false"):
        if low in ("true", "false"):
            return low == "true"
    if isinstance(current, int) and not isinstance(current, bool):
        try:
            return int(text)
        except ValueError:
            pass
    if isinstance(current, float):
        return float(text)
    if isinstance(current, list):
        return [t.strip().strip("'\"") for t in text.strip("[]").split(",") if t.strip()]
    if current is None:
        for cast in (int, float):
            try:
                return cast(text)
            except ValueError:
                pass
    return text


def apply_overrides(cfg: RunConfig, argv):
    """``++a.b=c`` / ``+a.b=c`` / ``a.b=c`` overrides; ``--local_rank=N`` (DeepSpeed launcher,
    Multitask/utils/deepspeed_utils.py:79-81) and ``hydra.*`` keys are accepted and ignored."""
    for arg in argv:
        if arg.startswith("--local_rank") or arg.startswith("hydra."):
            continue
        if "=" not in arg:
            raise ValueError(f"cannot parse override {arg!r} (expected ++section.key=value)")
        key, val = arg.lstrip("+").split("=", 1)
        parts = key.split(".")
        obj = cfg
        for p in parts[:-1]:
            if not hasattr(obj, p):
                raise KeyError(f"unknown config section {p!r} in {arg!r}")
            obj = getattr(obj, p)
        leaf = parts[-1]
        setattr(obj, leaf, _coerce(val, getattr(obj, leaf, None)))   # '++' may add new keys, like hydra
    return cfg


def parse_args(argv):
    return apply_overrides(RunConfig(), argv)


def load_ds_config(path_or_dict):
    """Multitask/conf/ds_config.json -> the numbers the engine needs (optimizer, schedule, accumulation)."""
    d = path_or_dict if isinstance(path_or_dict, dict) else json.load(open(path_or_dict))
    opt = d.get("optimizer", {}).get("params", {})
    sch = d.get("scheduler", {}).get("params", {})
    return dict(lr=float(opt.get("lr", 5e-5)), betas=tuple(opt.get("betas", (0.9, 0.999))), eps=float(opt.get("eps", 1e-6)),
                # DeepSpeed FusedAdam(adam_w_mode=True) default when ds_config omits it; torch.optim.AdamW's would be 0.01
                weight_decay=float(opt.get("weight_decay", 0.0)),
                warmup_num_steps=int(sch.get("warmup_num_steps", 200)), total_num_steps=int(sch.get("total_num_steps", 15000)),
                warmup_min_ratio=float(sch.get("warmup_min_ratio", 0.0)), cos_min_ratio=float(sch.get("cos_min_ratio", 1e-4)),
                warmup_type=sch.get("warmup_type", "log"),
                gradient_accumulation_steps=int(d.get("gradient_accumulation_steps", 1)),
                bf16=bool(d.get("bf16", {}).get("enabled", False)))


DEFAULT_DS_CONFIG = {
    "train_micro_batch_size_per_gpu": 1, "gradient_accumulation_steps": 1,
    "optimizer": {"type": "AdamW", "params": {"lr": 5e-5, "betas": [0.9, 0.999], "eps": 1e-6}},
    "scheduler": {"type": "WarmupCosineLR", "params": {"warmup_num_steps": 200, "total_num_steps": 15000}},
}
