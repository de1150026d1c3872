"""Decode entrypoint with the surface of the reference's ``Multitask/inference_batch.py`` (:44-151): build the model
through the plugin, loop ``model.generate(**batch)`` -> ``tokenizer.batch_decode`` -> write ``<key>\\t<text>`` lines to
``{decode_log}_pred`` / ``{decode_log}_gt``.

    python -m ps_slm_amd.inference_batch ++model_config.llm_path=synthetic:qwen2.5-1.5b \
        ++model_config.encoder_projector=linear-silu ++train_config.freeze_llm=true ++train_config.gt_emb=true \
        ++train_config.ctc_posterior=true ++dataset_config.file=synthetic ++decode_log=/tmp/decode_log
"""
import logging
import os
import random
import sys
import time

import torch

from .config import parse_args
from .finetune_deepspeed import get_custom_model_factory, get_dataset

logger = logging.getLogger(__name__)


def main(argv=None):
    from .streams import ensure_hw_queues
    ensure_hw_queues()                   # (before the first HIP call: ps_slm_amd/streams.py)
    cfg = parse_args(sys.argv[1:] if argv is None else argv)
    train_config, model_config, dataset_config = cfg.train_config, cfg.model_config, cfg.dataset_config
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s][%(name)s][%(levelname)s] - %(message)s")
    torch.manual_seed(train_config.seed)
    random.seed(train_config.seed)
    dev = f"cuda:{train_config.device or 0}"
    torch.cuda.set_device(dev)
    model_factory = get_custom_model_factory(model_config)
    ckpt = cfg.ckpt_path if cfg.ckpt_path and os.path.isfile(str(cfg.ckpt_path)) else None
    model, tokenizer = model_factory(train_config, model_config, ckpt_path=ckpt, metric=cfg.metric, device=dev,
                                     with_encoder=not train_config.gt_emb)
    model.eval()
    dataset = get_dataset(dataset_config, tokenizer, "test", model.core.geo, 0,
                          steps=int(cfg.get("synthetic_steps", 2)), batch_size=int(cfg.get("synthetic_batch", 4)))
    pred_path, gt_path = cfg.decode_log + "_pred", cfg.decode_log + "_gt"
    os.makedirs(os.path.dirname(os.path.abspath(pred_path)), exist_ok=True)
    n_tok, t0 = 0, time.perf_counter()
    with open(pred_path, "w") as pred, open(gt_path, "w") as gt:
        for raw in dataset:
            batch = dataset.collator(raw)
            keys, targets = batch.pop("keys"), batch.pop("targets")
            batch.pop("GT", None)
            out = model.generate(**batch, targets=targets, max_new_tokens=int(cfg.get("max_new_tokens", 200)))
            n_tok += out.numel()
            texts = model.tokenizer.batch_decode(out, add_special_tokens=False, skip_special_tokens=True)
            for key, text, target in zip(keys, texts, targets):
                pred.write(key + "\t" + text.replace("\n", " ") + "\n")
                gt.write(key + "\t" + target + "\n")
    logger.info("decoded %d tokens in %.2f s", n_tok, time.perf_counter() - t0)
    return pred_path, gt_path


if __name__ == "__main__":
    main()
