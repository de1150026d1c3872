"""A/B of the Qwen2.5-7B decode step on the weight-streaming kernels (K = 3584 in one range on row halves, the down projection's
K = 18944 as 12 x 1536 + 512 slabs; round 5) against the split-K kernels + finish launches it ran on before: ms per generated
position, alternating runs in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.config import ModelConfig, TrainConfig
from ps_slm_amd.ps_slm import model_factory
from ps_slm_amd.synthetic import synthetic_text_batch
from ps_slm_amd.decode import beam_search_generate

tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, gt_emb_noise=False, ctc_posterior=True, do_psd=True)
mc = ModelConfig(llm_path="synthetic:qwen2.5-7b", encoder_projector="linear-silu", encoder_dim=25055, llm_dim=3584)
model, _ = model_factory(tc, mc, device="cuda:0", init_seed=1234, keep_logits=False)
core = model.core
raw = synthetic_text_batch(core.geo, 16, seed=1234, noise=False)
ids = raw["input_ids"][:, :25]
am = torch.ones_like(ids, dtype=torch.bool)
NEW = 96


def run(stream, new, prenorm=True):
    core.ops.dec_stream_7b = stream
    core.ops.dec_prenorm = prenorm
    core._dec_graphs.clear(); core._dec_seen.clear()
    st = core.prepare_text(ids, am, None, raw["post_ids"], None, None)
    core.forward_projector_text(st)
    return beam_search_generate(core, st, num_beams=4, max_new_tokens=new, eos_token_id=-1, pad_token_id=0)


def timed(stream, new, prenorm=True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = run(stream, new, prenorm)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, o


outs = {s: run(s, NEW) for s in (True, False)}
same = (outs[True] == outs[False]).float().mean().item()
for rep in range(3):
    for s in (True, False):
        # the step alone: (time of NEW positions - time of NEW / 2 positions) / (NEW / 2) takes the prefill and set-up out
        t_full, _ = timed(s, NEW)
        t_half, _ = timed(s, NEW // 2)
        print(f"{'streaming' if s else 'split-K  '}: {(t_full - t_half) / (NEW - NEW // 2) * 1e3:.3f} ms/position "
              f"(whole call {t_full / NEW * 1e3:.3f} ms/position incl. prefill)   tokens equal to the other path: {same:.3f}", flush=True)

# the post-attention norm inside its neighbours (tasu_gemm_stream_resid_prenorm / _swiglu_rstd) at this geometry
for rep in range(3):
    for pn in (True, False):
        t_full, _ = timed(True, NEW, pn)
        t_half, _ = timed(True, NEW // 2, pn)
        print(f"streaming, prenorm {'on ' if pn else 'off'}: {(t_full - t_half) / (NEW - NEW // 2) * 1e3:.3f} ms/position", flush=True)
