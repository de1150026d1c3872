import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # as the entrypoints do (ps_slm_amd/streams.py:ensure_hw_queues): before the first HIP call

import numpy as np
import pytest
import torch


def free_port():
    """A TCP port nobody listens on right now (bind to 0, read it back): rendezvous ports derived from the pid collided between
    a module fixture's group and a test's own two-rank group once in a while (EADDRINUSE)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def split_flat(flat, lens):
    out, o = [], 0
    for n in lens.tolist():
        out.append(flat[o:o + n].tolist())
        o += n
    return out


@pytest.fixture(scope="session")
def geo():
    g = load_npz("geometry")
    return {k: (float(v) if k == "rope_theta" else int(v)) for k, v in g.items()}


@pytest.fixture(scope="session")
def tiny_weights():
    w = load_npz("weights_tiny")
    return {k: torch.from_numpy(v) for k, v in w.items()}


def golden_batch(name):
    """Fixture npz -> (batch dict of torch tensors in the oracle's schema, raw dict)."""
    z = load_npz(name)
    b = {}
    for k in ("input_ids", "attention_mask", "labels", "input_features", "input_feature_length"):
        if k in z:
            b[k] = torch.from_numpy(z[k])
    if "post_ids_flat" in z:
        b["post_ids"] = split_flat(z["post_ids_flat"], z["post_lens"])
    if "alphas" in z:
        b["alphas"] = z["alphas"].tolist()
        b["keeps"] = split_flat(z["keeps_flat"], z["post_lens"])
    return b, z


def mid_audio_psd_case():
    """(geo, state dict, batch, fixture) of tests/golden/mid_audio_psd.npz (see oracle/make_golden.py:main_mid)."""
    from ps_slm_amd.model import Geometry
    from ps_slm_amd.synthetic import MID_GEOMETRY, random_state_dict, synthetic_text_batch

    z = load_npz("mid_audio_psd")
    geo = Geometry.from_dict(MID_GEOMETRY)
    sd = random_state_dict(geo, int(z["seed_w"]), with_encoder=True)
    sd["encoder.ctc.ctc_lo.weight"] = sd["encoder.ctc.ctc_lo.weight"] * float(z["ctc_weight_scale"])
    sd["encoder.ctc.ctc_lo.bias"] = torch.from_numpy(z["ctc_bias"])
    batch = synthetic_text_batch(geo, 3, seed=int(z["seed_b"]), prompt_len=9, n_audio=21, target_len=17, speech_pos=4,
                                 feat_frames=12, noise=True, drop_prob=0.15, ragged=True)
    batch["input_features"] = torch.from_numpy(z["input_features"]).float()
    batch["input_feature_length"] = torch.from_numpy(z["input_feature_length"])
    return geo, sd, batch, z


def decode_margin_cases():
    """(geo, state dict, cases) of tests/golden/mid_generate_margin.npz (oracle/make_golden_generate_margin.py): decode cases
    whose beam-search decisions are stable under bf16 rounding noise, so token ids are compared EXACTLY.  Each case:
    dict(ids, am, post_ids, kw, tokens) with kw = generate() keyword arguments and tokens = the REAL reference's output."""
    from ps_slm_amd.model import Geometry
    from ps_slm_amd.synthetic import MID_GEOMETRY, decode_fixture_state_dict

    z = load_npz("mid_generate_margin")
    # the generator COUNTS stable cases its CPU double (the product's host code) got wrong instead of dropping them
    assert int(z["double_disagreements"]) == 0
    geo = Geometry.from_dict(MID_GEOMETRY)
    sd = decode_fixture_state_dict(geo, int(z["seed_w"]))
    cases = []
    for n in range(int(z["n_cases"])):
        nb, new, min_len = (int(v) for v in z[f"c{n}_kw"])
        cases.append(dict(ids=torch.from_numpy(z[f"c{n}_input_ids"]), am=torch.from_numpy(z[f"c{n}_attention_mask"]),
                          post_ids=split_flat(z[f"c{n}_post_ids_flat"], z[f"c{n}_post_lens"]), tokens=z[f"c{n}_tokens"],
                          kw=dict(num_beams=nb, max_new_tokens=new, min_length=min_len,
                                  length_penalty=float(z[f"c{n}_length_penalty"]))))
    return geo, sd, cases


def decode_fp32_cases():
    """(geo, state dict, cases, bf16_oracle_agrees) of tests/golden/mid_generate_fp32.npz (oracle/make_golden_generate_fp32.py):
    24 UNFILTERED random decode cases with the REAL reference's fp32 ``generate`` tokens -- every prompt the seeded generator drew,
    rounding-sensitive or not (the fp32 decode path is pinned on them; 8 of the 24 decode differently under bf16 rounding)."""
    from ps_slm_amd.model import Geometry
    from ps_slm_amd.synthetic import MID_GEOMETRY, decode_fixture_state_dict

    z = load_npz("mid_generate_fp32")
    geo = Geometry.from_dict(MID_GEOMETRY)
    sd = decode_fixture_state_dict(geo, int(z["seed_w"]))
    cases = []
    for n in range(int(z["n_cases"])):
        nb, new, min_len = (int(v) for v in z[f"c{n}_kw"])
        cases.append(dict(ids=torch.from_numpy(z[f"c{n}_input_ids"]), am=torch.from_numpy(z[f"c{n}_attention_mask"]),
                          post_ids=split_flat(z[f"c{n}_post_ids_flat"], z[f"c{n}_post_lens"]), tokens=z[f"c{n}_tokens"],
                          kw=dict(num_beams=nb, max_new_tokens=new, min_length=min_len,
                                  length_penalty=float(z[f"c{n}_length_penalty"]))))
    return geo, sd, cases, [bool(v) for v in z["bf16_oracle_agrees"]]


def decode_lora_margin_cases():
    """(geo, LoraConfig, state dict, adapter state dict, cases) of tests/golden/mid_generate_lora_margin.npz
    (oracle/make_golden_generate_lora_margin.py): rounding-stable decode cases of the LoRA-ADAPTED model.  tokens = the REAL
    reference's generate() with the LoRA formula applied by hand; tokens_base = what the un-adapted model decodes (different)."""
    from ps_slm_amd.lora import LoraConfig
    from ps_slm_amd.model import Geometry
    from ps_slm_amd.synthetic import MID_GEOMETRY, decode_fixture_state_dict, random_lora_state_dict

    z = load_npz("mid_generate_lora_margin")
    assert int(z["double_disagreements"]) == 0
    geo = Geometry.from_dict(MID_GEOMETRY)
    cfg = LoraConfig(r=int(z["r"]), lora_alpha=float(z["alpha"]), lora_dropout=0.0, target_modules=tuple(str(z["targets"]).split(",")))
    sd = decode_fixture_state_dict(geo, int(z["seed_w"]))
    lsd = random_lora_state_dict(geo, cfg, int(z["seed_l"]), b_scale=float(z["b_scale"]))
    cases = []
    for n in range(int(z["n_cases"])):
        nb, new, min_len = (int(v) for v in z[f"c{n}_kw"])
        cases.append(dict(ids=torch.from_numpy(z[f"c{n}_input_ids"]), am=torch.from_numpy(z[f"c{n}_attention_mask"]),
                          post_ids=split_flat(z[f"c{n}_post_ids_flat"], z[f"c{n}_post_lens"]), tokens=z[f"c{n}_tokens"],
                          tokens_base=z[f"c{n}_tokens_base"],
                          kw=dict(num_beams=nb, max_new_tokens=new, min_length=min_len,
                                  length_penalty=float(z[f"c{n}_length_penalty"]))))
    return geo, cfg, sd, lsd, cases


def ca_projector_case():
    """(geo, state dict, batch, fixture) of tests/golden/mid512_text_ca.npz (oracle/make_golden_ca.py): the alternate
    ``encoder_projector="cross-attention"`` (EncoderProjectorCTCCA) at llm_dim 512 (8 heads of 64)."""
    from ps_slm_amd.model import Geometry
    from ps_slm_amd.synthetic import MID_GEOMETRY, random_state_dict, synthetic_text_batch

    z = load_npz("mid512_text_ca")
    geo = Geometry.from_dict(dict(MID_GEOMETRY, projector="cross-attention", llm_dim=512, llm_heads=4, llm_kv_heads=2, llm_inter=1024))
    sd = random_state_dict(geo, int(z["seed_w"]), with_encoder=False)
    batch = synthetic_text_batch(geo, 3, seed=int(z["seed_b"]), prompt_len=9, n_audio=22, target_len=17, speech_pos=4,
                                 feat_frames=12, noise=False, ragged=True)
    return geo, sd, batch, z


def cov1d_projector_case(k):
    """(geo, state dict, batch, fixture) of tests/golden/mid_text_cov1d_k{k}.npz (oracle/make_golden_cov1d.py): the alternate
    ``encoder_projector="cov1d-linear"`` (EncoderProjectorCov1d: Conv1d kernel = stride = k -> ReLU -> Linear -> ReLU -> Linear)."""
    from ps_slm_amd.model import Geometry
    from ps_slm_amd.synthetic import MID_GEOMETRY, random_state_dict, synthetic_text_batch

    z = load_npz(f"mid_text_cov1d_k{k}")
    geo = Geometry.from_dict(dict(MID_GEOMETRY, projector="cov1d-linear", projector_ds_rate=k, bottleneck=2048))
    sd = random_state_dict(geo, int(z["seed_w"]), with_encoder=False)
    batch = synthetic_text_batch(geo, 3, seed=int(z["seed_b"]), prompt_len=9, n_audio=22, target_len=17, speech_pos=4,
                                 feat_frames=12, noise=False, ragged=True)
    return geo, sd, batch, z


def linear_projector_case(k):
    """(geo, state dict, batch, fixture) of tests/golden/mid_text_linear_k{k}.npz (oracle/make_golden_linear.py): the alternate
    ``encoder_projector="linear"`` (EncoderProjectorConcat) with k frames concatenated per projector row."""
    from ps_slm_amd.model import Geometry
    from ps_slm_amd.synthetic import MID_GEOMETRY, random_state_dict, synthetic_text_batch

    z = load_npz(f"mid_text_linear_k{k}")
    geo = Geometry.from_dict(dict(MID_GEOMETRY, projector="linear", projector_ds_rate=k, bottleneck=2048))
    sd = random_state_dict(geo, int(z["seed_w"]), with_encoder=False)
    batch = synthetic_text_batch(geo, 3, seed=int(z["seed_b"]), prompt_len=9, n_audio=22, target_len=17, speech_pos=4,
                                 feat_frames=12, noise=False, ragged=True)
    return geo, sd, batch, z


def mid_audio_raw_case(kind, k):
    """The inputs of tests/golden/mid_audio_raw_<kind>_k<k>.npz (oracle/make_golden_raw.py): the encoder, CTC head and features
    of mid_audio_psd (PSD decisions stable under bf16 rounding) with a projector that reads the encoder's output states
    (train_config.ctc_posterior=false): (geo, state dict, batch, fixture)."""
    import dataclasses

    from ps_slm_amd.model import Geometry
    from ps_slm_amd.synthetic import random_state_dict

    geo0, sd, batch, _ = mid_audio_psd_case()
    z = load_npz(f"mid_audio_raw_{kind}_k{k}")
    geo = dataclasses.replace(geo0, projector=kind, projector_ds_rate=k, proj_in=geo0.enc_dim,
                              bottleneck=2048 if kind == "linear" else geo0.bottleneck)
    sd = {n: v for n, v in sd.items() if not n.startswith("encoder_projector.")}
    sd.update({n: v for n, v in random_state_dict(geo, int(z["seed_p"]), with_encoder=False).items() if n.startswith("encoder_projector.")})
    return geo, sd, batch, z
