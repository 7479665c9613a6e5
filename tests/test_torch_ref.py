"""CPU test that PINS the dropout checker (``tests/torch_ref.py``) to the imported reference.

``tests/test_hip_train.py`` and ``tools/fuzz_train.py`` compare the HIP training path with ``torch_ref.forward_with_masks``
(a float64 restatement with explicit dropout masks).  That restatement is a builder artefact: this test holds it, with
the masks off, to the float64 gradients the IMPORTED reference produced (``tests/golden/make_golden_train.py``:
reference ``SimNet`` in train mode, dropout 0, + reference ``utils.mse_with_mask_loss``; ``src/model/simnet.py:32-45``,
``src/utils/utils.py:45-56``), for every ``tests/golden/train_*.npz``.

Tolerances.  The goldens store the float64 LOSS and per-tensor float64 SUM / L2 NORM exactly, and the sampled gradient
rows rounded to fp32.  So: loss and whole-tensor sum / norm to 1e-10 relative (two float64 runs that differ only in
summation order), sampled rows to fp32 storage rounding (6e-8 of the value + 1e-12 of the tensor's maximum)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
import torch_ref


def _cases():
    with open(os.path.join(GOLDEN, "train_index.json")) as f:
        return json.load(f)["cases"]


def _inputs(synth, c):
    x = synth.make_features(c["B"], c["T"], c["xseed"], c["kind"], c.get("lengths"))
    mask = None
    if c.get("lengths") is not None:
        mask = synth.padding_mask(x)
    if c.get("randmask") is not None:
        mask = synth.random_mask(c["B"], c["T"], c["randmask"])
    rng = np.random.Generator(np.random.PCG64(c["tseed"]))
    target = torch.from_numpy(rng.random(size=(c["B"], c["T"])).astype(np.float32))
    R = torch.from_numpy(rng.standard_normal(size=(c["B"], c["T"], c["d"])).astype(np.float32))
    return x, mask, target, R


def mse_with_mask_loss_f64(output, targets, mask):
    """reference utils.py:45-56, reduction 'avg': masked frames are scaled by 0, the mean runs over ALL B*T entries"""
    scale = torch.ones_like(targets)
    scale = scale.masked_fill(mask, 0.0)
    return (((output.squeeze(2) - targets) * scale) ** 2).mean()


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_torch_ref_without_masks_reproduces_the_reference_float64_gradients(vsa, case):
    c = case
    z = np.load(os.path.join(GOLDEN, c["name"] + ".npz"))
    sd = vsa.synth.make_state_dict(c["d"], c["L"], c["wseed"])
    x, mask, target, R = _inputs(vsa.synth, c)
    params = {k: v.double().clone().requires_grad_(v.dtype.is_floating_point and "pos_embedding" not in k)
              for k, v in sd.items()}
    xx = x.double().clone().requires_grad_(True)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    pred, hidden = torch_ref.forward_with_masks(params, xx, mask, c["H"], p=0.0, p_embed=0.0, masks=None)
    mk = mask if mask is not None else torch.zeros(x.shape[:2], dtype=torch.bool)
    loss = mse_with_mask_loss_f64(pred, target.double(), mk)
    if c["hidden_w"]:
        loss = loss + c["hidden_w"] * (hidden * R.double()).sum()
    loss.backward()

    want_loss = float(z["loss"])
    assert abs(loss.item() - want_loss) <= 1e-12 * max(1.0, abs(want_loss)), (loss.item(), want_loss)
    assert (pred.detach().float() - torch.from_numpy(z["logits"])).abs().max().item() <= 1e-6

    grads = {"x": xx.grad}
    grads.update({k: p.grad for k, p in params.items() if p.requires_grad})
    keys = json.loads(str(z["keys"]))
    assert sorted(keys) == sorted(grads.keys())
    worst = 0.0
    for k in keys:
        g = grads[k]
        assert g is not None, k
        tot, nrm, gmax, _ = z["s:" + k]
        g2 = g.reshape(-1, g.shape[-1]) if g.dim() > 1 else g.reshape(1, -1)
        rows = torch.from_numpy(z["r:" + k])
        want = torch.from_numpy(z["g:" + k]).double()
        got = g2[rows]
        err = (got - want).abs()
        bound = 6.1e-8 * want.abs() + 1e-12 * gmax + 1e-300         # fp32 storage rounding of the golden
        assert bool((err <= bound).all()), "%s: max err %.3e (max |g| %.3e)" % (k, err.max().item(), gmax)
        assert abs(g.sum().item() - tot) <= 1e-10 * max(abs(tot), nrm) + 1e-300, k
        assert abs(g.norm().item() - nrm) <= 1e-10 * nrm + 1e-300, k
        worst = max(worst, abs(g.norm().item() - nrm) / (nrm + 1e-300))
    print("%s: torch_ref vs reference float64: worst relative norm difference %.1e" % (c["name"], worst))


def test_torch_ref_explicit_all_ones_masks_equal_no_masks(vsa):
    """keep masks of all ones with p = 0 must be the identity (the explicit-mask plumbing itself)"""
    sd = vsa.synth.make_state_dict(256, 1, 3)
    x = vsa.synth.make_features(1, 40, 5, "randn", None).double()
    params = {k: v.double() for k, v in sd.items()}
    B, T, d, H = 1, 40, 256, 4
    ones = dict(embed=torch.ones(B, T, d, dtype=torch.bool), attn0=torch.ones(B, H, T, T, dtype=torch.bool),
                drop1_0=torch.ones(B, T, d, dtype=torch.bool), mlp0=torch.ones(B, T, 4 * d, dtype=torch.bool),
                drop2_0=torch.ones(B, T, d, dtype=torch.bool))
    a, _ = torch_ref.forward_with_masks(params, x, None, H)
    b, _ = torch_ref.forward_with_masks(params, x, None, H, p=0.0, p_embed=0.0, masks=ones)
    assert torch.equal(a, b)
