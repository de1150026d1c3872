"""Decode leg of bench.py alone (beam 4, 16 utterances, prefill 128, 200 generated positions): python tools/bench_decode.py
[--model qwen2.5-1.5b] [--batch 16].  HipOps.use_stream = False selects the split-K launch scheme (A/B runs)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ps_slm_amd.config import ModelConfig, TrainConfig
from ps_slm_amd.ps_slm import model_factory
from ps_slm_amd.synthetic import synthetic_text_batch

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="qwen2.5-1.5b")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--new-tokens", type=int, default=200)
a = ap.parse_args()
tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, gt_emb_noise=True, ctc_posterior=True, do_psd=True,
                 use_fp16=True, batching_strategy="dynamic")
mc = ModelConfig(llm_path=f"synthetic:{a.model}", encoder_projector="linear-silu", encoder_dim=25055,
                 llm_dim={"qwen2.5-1.5b": 1536, "qwen2.5-7b": 3584, "mid": 256}[a.model])
model, _ = model_factory(tc, mc, device="cuda:0", init_seed=1234, keep_logits=False, with_encoder=False)
raw = synthetic_text_batch(model.core.geo, a.batch, seed=1234, noise=False)
print(json.dumps(bench.decode_leg(model.core, raw, a.batch, new_tokens=a.new_tokens)))
