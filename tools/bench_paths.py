"""Secondary measurements at full Qwen2.5-1.5B geometry on one MI355X: decode tok/s (beam 4) and the audio-SFT step
(real features through the SenseVoice encoder).  Prints one JSON object per measurement."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ps_slm_amd.config import DEFAULT_DS_CONFIG, ModelConfig, TrainConfig, load_ds_config
from ps_slm_amd.engine import TasuEngine
from ps_slm_amd.ps_slm import model_factory
from ps_slm_amd.synthetic import synthetic_text_batch
from ps_slm_amd.decode import beam_search_generate

what = sys.argv[1] if len(sys.argv) > 1 else "decode"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
MODEL = sys.argv[3] if len(sys.argv) > 3 else "qwen2.5-1.5b"          # or qwen2.5-7b
tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=(what == "decode"), gt_emb_noise=False, ctc_posterior=True, do_psd=True)
mc = ModelConfig(llm_path=f"synthetic:{MODEL}", encoder_projector="linear-silu", encoder_dim=25055, llm_dim={"qwen2.5-1.5b": 1536, "qwen2.5-7b": 3584}[MODEL])
model, tok = model_factory(tc, mc, device="cuda:0", init_seed=1234, keep_logits=False, with_encoder=(what != "decode"))
core = model.core
geo = core.geo
raw = synthetic_text_batch(geo, B, seed=1234, noise=False)
if what == "decode":
    # prompt without targets: 24 prompt ids + <speech> -> S_p = 24 + 104 = 128, beams 4, 200 forced new tokens
    ids = raw["input_ids"][:, :25]
    am = torch.ones_like(ids, dtype=torch.bool)
    new = 200 if MODEL == "qwen2.5-1.5b" else 64
    def run():
        st = core.prepare_text(ids, am, None, raw["post_ids"], None, None)
        core.forward_projector_text(st)
        # eos = -1 never matches: every run emits exactly B x 200 tokens (min_length = max_new_tokens in the plan)
        return beam_search_generate(core, st, num_beams=4, max_new_tokens=new, eos_token_id=-1, pad_token_id=0)
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"what": "decode", "B": B, "beams": 4, "prefill_len": 128, "new_tokens": int(out.shape[1]),
                      "seconds": round(dt, 3), "emitted_tok_per_s": round(B * out.shape[1] / dt, 1),
                      "beam_tok_per_s": round(4 * B * out.shape[1] / dt, 1), "ms_per_step": round(dt / out.shape[1] * 1e3, 3),
                      "weights_bytes_per_step": 2 * 1543714304}))
else:
    core.use_graphs = True
    eng = TasuEngine(model, load_ds_config(DEFAULT_DS_CONFIG))
    eng.train()
    batch = dict(input_ids=raw["input_ids"], attention_mask=raw["attention_mask"], labels=raw["labels"],
                 input_features=raw["input_features"], input_feature_length=raw["input_feature_length"], GT=None)
    def step():
        out, acc = eng(**batch)
        eng.backward(out.loss)
        eng.step()
        return out
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    st = eng._last_state
    print(json.dumps({"what": "audio_sft_step", "B": B, "encoder_frames": 504, "psd_audio_tokens_mean": float(np.mean(st.dev["psd_lens"])),
                      "S": st.S, "ms_per_step": round(dt * 1e3, 2), "utt_per_s": round(B / dt, 1), "loss": float(out.loss.detach())}))
