import os
import sys

import numpy as np
import pytest
import torch

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
