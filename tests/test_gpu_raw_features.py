"""The raw-feature branch (ctc_posterior=false) on the MI355X: HIP kernels against the reference goldens and the CPU double."""
import pytest
import torch

from conftest import mid_audio_raw_case
from fake_ops import FakeOps
from ps_slm_amd.model import TasuModel
from test_raw_features_cpu import CASES, check, cosine, run_audio

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,k", CASES)
def test_raw_branch_hip_vs_double_and_reference(kind, k):
    from ps_slm_amd.ops import HipOps
    geo, sd, batch, z = mid_audio_raw_case(kind, k)
    gm = TasuModel(geo, HipOps(), "cuda")
    gm.load_reference_state_dict(sd)
    cm = TasuModel(geo, FakeOps(), "cpu")
    cm.load_reference_state_dict(sd)
    sg, sc = run_audio(gm, batch), run_audio(cm, batch)
    torch.cuda.synchronize()
    assert abs(float(sg.dev["loss_out"][0]) - float(sc.dev["loss_out"][0])) < 3e-3
    assert sg.S == sc.S and sg.Ra == sc.Ra
    for name, g in cm.projector_grads().items():
        assert cosine(gm.projector_grads()[name], g) > 0.999, name
    check(gm, sg, z)
    # beam-4 generate() through the same front end: the double's tokens on a long common prefix
    from ps_slm_amd.decode import beam_search_generate
    ids, am = batch["input_ids"][:, :9], batch["attention_mask"][:, :9]
    outs = []
    for m in (gm, cm):
        st = m.prepare_audio(ids, am, None, batch["input_features"], batch["input_feature_length"])
        outs.append(beam_search_generate(m, st, max_new_tokens=10).numpy())
    assert ((outs[0] == outs[1]).cumprod(1).sum(1) >= 4).all(), outs
