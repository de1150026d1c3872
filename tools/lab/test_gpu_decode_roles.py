"""The decoder layers of a generated position as ONE launch (csrc/decode_roles.hip: seven roles per layer in one grid, weight
tiles requested before the dependency wait) against the one-launch-per-GEMM decode step: the same bodies in another launch
structure must give the SAME BITS -- token ids, beam scores, back-pointers and the K / V cache -- at a small geometry that
takes every role (hidden 256, 2 q / 1 kv heads, down projection in 5 K-range slabs) and at Qwen2.5-1.5B's full geometry with the
benchmark's 16 utterances x 4 beams; 48 rows (the last 16-row tile partly empty); a context that straddles cache blocks."""
import pytest
import torch

from ps_slm_amd.decode import beam_search_generate
from ps_slm_amd.model import Geometry, TasuModel
from ps_slm_amd.synthetic import synthetic_text_batch

pytestmark = pytest.mark.gpu

SMALL = dict(llm_vocab=1000, llm_dim=256, llm_inter=1280, llm_layers=3, llm_heads=2, llm_kv_heads=1, rope_theta=1e6, tied=True,
             ctc_vocab=203, bottleneck=128, feat_dim=80, enc_dim=256, enc_heads=2, enc_ffn=512, enc_blocks=2, enc_tp_blocks=1,
             enc_kernel=11, speech_id=990, eos_id=980)


def run(m, batch, B, nb, n_new, roles, prompt=25, **kw):
    m.ops.dec_roles = roles
    ids = batch["input_ids"][:B, :prompt]
    st = m.prepare_text(ids, torch.ones_like(ids, dtype=torch.bool), None, batch["post_ids"][:B], None, None)
    m.forward_projector_text(st)
    out = beam_search_generate(m, st, num_beams=nb, max_new_tokens=n_new, pad_token_id=0, **kw)
    torch.cuda.synchronize()
    bs = m._last_beam
    state = [t.clone() for t in (bs.fin_scores, bs.run_scores, bs.bp_tok, bs.bp_par, bs.fin_len, bs.fin_tok)]
    cache = [m._ws["dec_kc"].clone(), m._ws["dec_vc"].clone()]
    took = getattr(m, "_dec_took_roles", None)
    return out, state, cache, took


@pytest.mark.parametrize("B,nb,n_new", [(16, 4, 12), (12, 4, 9), (13, 3, 40)])
def test_roles_launch_equals_per_gemm_launches_small_geometry(B, nb, n_new):
    from ps_slm_amd.ops import HipOps
    geo = Geometry.from_dict(SMALL)
    m = TasuModel(geo, HipOps(), "cuda", keep_logits=False)
    m.init_random(seed=77)
    batch = synthetic_text_batch(geo, 16, seed=5, noise=False, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=8)
    ref = run(m, batch, B, nb, n_new, roles=False, prompt=9, eos_token_id=geo.eos_id)
    got = run(m, batch, B, nb, n_new, roles=True, prompt=9, eos_token_id=geo.eos_id)
    assert ref[3] is False and got[3] is True                     # the second run really took the one-launch path
    assert torch.equal(ref[0], got[0])
    for a, b in zip(ref[1], got[1]):
        assert torch.equal(a, b)
    for a, b in zip(ref[2], got[2]):
        assert torch.equal(a, b)
    got2 = run(m, batch, B, nb, n_new, roles=True, prompt=9, eos_token_id=geo.eos_id)      # graph replay of the same shape
    assert torch.equal(got[0], got2[0]) and all(torch.equal(a, b) for a, b in zip(got[1], got2[1]))


def test_roles_launch_equals_per_gemm_launches_at_qwen25_1p5b():
    from test_gpu_fullsize import full_model
    geo, m = full_model("1.5b")
    batch = synthetic_text_batch(geo, 16, seed=321, noise=False)
    try:
        ref = run(m, batch, 16, 4, 10, roles=False, eos_token_id=-1)
        got = run(m, batch, 16, 4, 10, roles=True, eos_token_id=-1)
    finally:
        m.ops.dec_roles = True
    assert ref[3] is False and got[3] is True
    assert torch.equal(ref[0], got[0])
    for a, b in zip(ref[1], got[1]):
        assert torch.equal(a, b)
    for a, b in zip(ref[2], got[2]):
        assert torch.equal(a, b)
