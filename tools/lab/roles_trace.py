"""Per-role timeline of one decoder layer inside the one-launch decode step (instrumented build: make -C ps_slm_amd/csrc trace;
TASU_LIB_PATH=ps_slm_amd/libtasu_hip_trace.so): for every workgroup of layer 10, wall-clock stamps at entry, after its dependency
wait, before its signal and at exit -> per role: first entry, last entry, first / last end of wait, first / last exit (us after the
layer's first entry), averaged over the positions of one generate()."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ps_slm_amd.config import ModelConfig, TrainConfig
from ps_slm_amd.ps_slm import model_factory
from ps_slm_amd.synthetic import synthetic_text_batch
from ps_slm_amd.decode import beam_search_generate

tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, gt_emb_noise=False, ctc_posterior=True, do_psd=True)
mc = ModelConfig(llm_path="synthetic:qwen2.5-1.5b", encoder_projector="linear-silu", encoder_dim=25055, llm_dim=1536)
model, tok = model_factory(tc, mc, device="cuda:0", init_seed=1234, keep_logits=False)
core = model.core
core.decode_graphs = False                                   # eager: the stamps of the LAST position are what is read
raw = synthetic_text_batch(core.geo, 16, seed=1234, noise=False)
ids = raw["input_ids"][:, :25]
st = core.prepare_text(ids, torch.ones_like(ids, dtype=torch.bool), None, raw["post_ids"], None, None)
core.forward_projector_text(st)
beam_search_generate(core, st, num_beams=4, max_new_tokens=int(sys.argv[1]) if len(sys.argv) > 1 else 100, eos_token_id=-1, pad_token_id=0)
torch.cuda.synchronize()
lib = core.ops.lib
lib.tasu_roles_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
cum = core._dec_roles[2].numpy()
per = int(cum[7])
buf = (ctypes.c_uint64 * (4 * per))()
assert lib.tasu_roles_trace_read(buf, 4 * per) == 0
t = np.array(buf[:], dtype=np.float64).reshape(per, 4) / 100.0
ok = t[:, 3] > 0
t0 = t[ok, 0].min()
names = ["qkv", "attn", "o", "norm", "gate|up", "down", "finish"]
print(f"layer 10 of the last position, {int(ok.sum())} workgroups with work of {per} blocks; us after the layer's first entry")
print(f"{'role':8s} {'wgs':>4s} {'entry first':>11s} {'entry last':>10s} {'waited first':>12s} {'waited last':>11s} {'exit first':>10s} {'exit last':>9s}  {'wait avg':>8s} {'work avg':>8s} {'signal avg':>10s}")
for k in range(7):
    sel = np.zeros(per, bool); sel[cum[k]:cum[k + 1]] = True; sel &= ok
    if not sel.any():
        continue
    a = t[sel] - t0
    print(f"{names[k]:8s} {int(sel.sum()):4d} {a[:,0].min():11.2f} {a[:,0].max():10.2f} {a[:,1].min():12.2f} {a[:,1].max():11.2f} {a[:,3].min():10.2f} {a[:,3].max():9.2f}  "
          f"{(a[:,1]-a[:,0]).mean():8.2f} {(a[:,2]-a[:,1]).mean():8.2f} {(a[:,3]-a[:,2]).mean():10.2f}")
