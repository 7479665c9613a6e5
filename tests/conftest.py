import importlib
import json
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


@pytest.fixture(scope="session")
def vsa():
    """The product package (hyphenated directory name -> importlib)."""
    return importlib.import_module("video-summarization_amd")


def golden_cases():
    with open(os.path.join(GOLDEN, "index.json")) as f:
        return json.load(f)["cases"]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = json.loads(str(z["cfg"]))
    mask = torch.from_numpy(z["mask"]) if z["mask"].size else None
    return dict(cfg=cfg, logits=torch.from_numpy(z["logits"]), hidden=torch.from_numpy(z["hidden"]),
                rows=torch.from_numpy(z["hidden_rows"]), mask=mask)


def build_case(synth, c):
    """Seeded weights + inputs of a golden case (same recipe as tests/golden/make_golden.py)."""
    sd = synth.make_state_dict(c["d"], c["L"], c["wseed"], num_classes=c.get("num_classes", 1),
                               use_pos=c.get("use_pos", True))
    x = synth.make_features(c["B"], c["T"], c["xseed"], c["kind"], c.get("lengths"))
    mask = None
    if c.get("lengths") is not None:
        mask = synth.padding_mask(x)
    if c.get("randmask") is not None:
        mask = synth.random_mask(c["B"], c["T"], c["randmask"])
    return sd, x, mask
