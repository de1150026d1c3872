"""A/B of the headline training step (1.5B, 16 x S = 256, hipGraph replay) with the attention kernel families forced, alternating
in ONE process (box-to-box noise is +-1 %, the differences looked for are smaller)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ps_slm_amd.streams import ensure_hw_queues
ensure_hw_queues()
import torch
from ps_slm_amd.config import DEFAULT_DS_CONFIG, ModelConfig, TrainConfig, load_ds_config
from ps_slm_amd.engine import TasuEngine
from ps_slm_amd.ps_slm import model_factory
from ps_slm_amd.synthetic import synthetic_text_batch

tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, gt_emb_noise=True, ctc_posterior=True, do_psd=True, use_fp16=True, batching_strategy="dynamic")
mc = ModelConfig(llm_path="synthetic:qwen2.5-1.5b", encoder_projector="linear-silu", encoder_dim=25055, llm_dim=1536)
model, _ = model_factory(tc, mc, device="cuda:0", init_seed=1234, keep_logits=False, with_encoder=False)
model.drop_prob = 0.0
core = model.core
core.use_graphs = True
engine = TasuEngine(model, load_ds_config(DEFAULT_DS_CONFIG))
engine.train()
raw = synthetic_text_batch(core.geo, 16, seed=1234, noise=False)
batch = dict(input_ids=raw["input_ids"], attention_mask=raw["attention_mask"], labels=raw["labels"], input_features=None, input_feature_length=None,
             GT=[" ".join(map(str, p)) for p in raw["post_ids"]])


def step():
    out, acc = engine(**batch)
    engine.backward(out.loss)
    engine.step()


def timed(n=20):
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


variants = [("fwd tiled, bwd per-head", "tiled", "tiled"), ("fwd tiled, bwd gqa (policy)", "tiled", "gqa"), ("fwd sp, bwd gqa", "sp", "gqa"),
            ("fwd tiled, bwd sp", "tiled", "sp")]
res = {v[0]: [] for v in variants}
for rnd in range(3):
    for name, f, b in variants:
        core.ops.attn_kernel = {"fwd": f, "bwd": b}
        core._graphs.clear(); core._graph_seen.clear()
        res[name].append(round(timed(), 3))
print(json.dumps({"ms_per_step": res}))
