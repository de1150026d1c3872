"""A/B of the decode down projection's K-range split (7 x 1280 with all 64 rows per workgroup vs 5 x 1792 with two row halves):
ms per generated position, alternating runs in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.config import ModelConfig, TrainConfig
from ps_slm_amd.ps_slm import model_factory
from ps_slm_amd.synthetic import synthetic_text_batch
from ps_slm_amd.decode import beam_search_generate

tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, gt_emb_noise=False, ctc_posterior=True, do_psd=True)
mc = ModelConfig(llm_path="synthetic:qwen2.5-1.5b", encoder_projector="linear-silu", encoder_dim=25055, llm_dim=1536)
model, _ = model_factory(tc, mc, device="cuda:0", init_seed=1234, keep_logits=False)
core = model.core
raw = synthetic_text_batch(core.geo, 16, seed=1234, noise=False)
ids = raw["input_ids"][:, :25]
am = torch.ones_like(ids, dtype=torch.bool)


def run(order):
    core.ops.dec_split_order = order
    core._dec_graphs.clear(); core._dec_seen.clear()
    st = core.prepare_text(ids, am, None, raw["post_ids"], None, None)
    core.forward_projector_text(st)
    return beam_search_generate(core, st, num_beams=4, max_new_tokens=200, eos_token_id=-1, pad_token_id=0)


A, Bo = (1, 7, 5, 2, 3, 4, 6, 8), (1, 5, 7, 2, 3, 4, 6, 8)
outs = {}
for name, order in (("7x1280", A), ("5x1792", Bo)):
    outs[name] = run(order)
torch.cuda.synchronize()
for rep in range(3):
    for name, order in (("7x1280", A), ("5x1792", Bo)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o = run(order)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name}: {dt / o.shape[1] * 1e3:.3f} ms/position  (same tokens as the other split: {bool((o == outs['5x1792']).all())})")
