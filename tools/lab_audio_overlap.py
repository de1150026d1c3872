"""Lab: how much of the frozen encoder pass of the audio-SFT step can hide under the previous batch's LLM step?
(1) the step as shipped; (2) the encoder graph alone; (3) the step with the encoder pass replaced by its cached output;
(4) = (3) with the encoder graph replayed on a side stream at the start of every step (upper bound of a one-batch-ahead pipeline)."""
import os, sys, time, json, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ps_slm_amd.encoder as encmod
from ps_slm_amd.config import DEFAULT_DS_CONFIG, ModelConfig, TrainConfig, load_ds_config
from ps_slm_amd.engine import TasuEngine
from ps_slm_amd.ps_slm import model_factory
from ps_slm_amd.synthetic import synthetic_text_batch

ap = argparse.ArgumentParser()
ap.add_argument("--bias", type=float, default=13.0)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--prio", type=int, nargs="*", default=[0, -1])
a = ap.parse_args()
B = 16
tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=False, gt_emb_noise=False, ctc_posterior=True, do_psd=True, use_fp16=True,
                 batching_strategy="dynamic")
mc = ModelConfig(llm_path="synthetic:qwen2.5-1.5b", encoder_projector="linear-silu", encoder_dim=25055, llm_dim=1536)
model, _ = model_factory(tc, mc, device="cuda:0", init_seed=1234, keep_logits=False, with_encoder=True)
core = model.core
core.use_graphs = True
engine = TasuEngine(model, load_ds_config(DEFAULT_DS_CONFIG))
engine.train()
geo = core.geo
raw = synthetic_text_batch(geo, B, seed=1234, noise=False)
batch = dict(input_ids=raw["input_ids"], attention_mask=raw["attention_mask"], labels=raw["labels"],
             input_features=raw["input_features"], input_feature_length=raw["input_feature_length"], GT=None)
core.encoder.ctc_b[geo.blank_id] += a.bias

def step():
    out, acc = engine(**batch)
    engine.backward(out.loss)
    engine.step()
    return out

def timeit(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for _ in range(4): step()
res = {"kept": engine._last_state.Ra / B, "S": engine._last_state.S}
res["step_ms"] = round(timeit(step, a.steps), 3)
real = encmod.encoder_posterior
feats, fl = batch["input_features"], batch["input_feature_length"]
res["encoder_alone_ms"] = round(timeit(lambda: real(core, feats, fl, want_post=False), a.steps), 3)
cached = real(core, feats, fl, want_post=False)
encmod.encoder_posterior = lambda m, f, l, want_post=True: cached
for _ in range(2): step()
res["step_without_encoder_ms"] = round(timeit(step, a.steps), 3)
for prio in a.prio:
    side = torch.cuda.Stream(priority=prio)
    def overlapped():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            real(core, feats, fl, want_post=False)
        step()
        torch.cuda.current_stream().wait_stream(side)
    for _ in range(2): overlapped()
    res[f"step_with_encoder_on_side_stream_prio{prio}_ms"] = round(timeit(overlapped, a.steps), 3)
# the reverse: the decoder step on a high-priority stream, the encoder on the default one
hi = torch.cuda.Stream(priority=-1)
def overlapped2():
    cur = torch.cuda.current_stream()
    hi.wait_stream(cur)
    real(core, feats, fl, want_post=False)
    with torch.cuda.stream(hi):
        step()
    cur.wait_stream(hi)
try:
    for _ in range(3): overlapped2()
    res["decoder_step_on_high_priority_stream_ms"] = round(timeit(overlapped2, a.steps), 3)
except Exception as e:
    res["decoder_step_on_high_priority_stream_ms"] = str(e)[:200]
print(json.dumps(res))
