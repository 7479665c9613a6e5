"""CPU: the oracle restatement against the reference-generated golden vectors."""
import pytest
import torch

from conftest import build_case, golden_cases, load_golden
from oracle.simnet_oracle import oracle_forward

TOL = 2e-5   # fp32 summation-order noise through L post-LN layers (survey probe: 6e-7 .. 1.5e-6)


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_oracle_matches_reference_golden(vsa, case):
    g = load_golden(case["name"])
    sd, x, mask = build_case(vsa.synth, case)
    if g["mask"] is not None:
        assert torch.equal(mask, g["mask"])
    with torch.no_grad():
        logits, hidden = oracle_forward(sd, x, mask, case["H"])
    assert logits.shape == g["logits"].shape
    assert (logits - g["logits"]).abs().max().item() < TOL
    assert (hidden[:, g["rows"]] - g["hidden"]).abs().max().item() < TOL


def test_oracle_fp64_noise_floor(vsa):
    """fp32 oracle vs the same restatement in fp64: the noise floor the 1e-4 bar sits above."""
    case = [c for c in golden_cases() if c["name"] == "ma_t320"][0]
    sd, x, mask = build_case(vsa.synth, case)
    with torch.no_grad():
        l32, _ = oracle_forward(sd, x, mask, case["H"])
        l64, _ = oracle_forward(sd, x, mask, case["H"], dtype=torch.float64)
    assert (l32.double() - l64).abs().max().item() < 1e-5


def test_padded_valid_frames_equal_unpadded(vsa):
    """with the key mask on, valid-frame outputs equal the unpadded run."""
    synth = vsa.synth
    sd = synth.make_state_dict(256, 2, 5)
    x = synth.make_features(1, 90, 77, "randn", [61])
    with torch.no_grad():
        lp, _ = oracle_forward(sd, x, synth.padding_mask(x), 4)
        lu, _ = oracle_forward(sd, x[:, :61], None, 4)
    assert (lp[:, :61] - lu).abs().max().item() < 1e-5


def test_oracle_class_token_branch_matches_reference_golden(vsa):
    """use_cls=True (simnet.py:205-206, 214-216): vectors from the imported reference (tests/golden/make_golden_cls.py)."""
    import json
    import os
    import numpy as np
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "cls_golden.npz"))
    for c in json.loads(str(z["cases"])):
        sd = vsa.synth.make_state_dict(c["d"], c["L"], c["wseed"], use_cls=True)
        x = vsa.synth.make_features(c["B"], c["T"], c["xseed"], "randn")
        logits, hidden = oracle_forward(sd, x, None, c["H"])
        assert logits.shape == (c["B"], c["T"] + 1, 1)
        assert (logits - torch.from_numpy(z[c["name"] + ":logits"])).abs().max().item() < 2e-5
        assert (hidden - torch.from_numpy(z[c["name"] + ":hidden"])).abs().max().item() < 2e-5
