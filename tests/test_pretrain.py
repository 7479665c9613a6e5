"""PretrainModel counterpart (video-summarization_amd/pretrain.py) against values produced by importing the
reference's PretrainModel (tests/golden/make_golden_pretrain.py).  GPU: the encoder under the head is the HIP
training path (forward + backward kernels); the golden values come from the reference on CPU."""
import importlib.util
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
DEV = torch.device("cuda:0")


def _maker():
    spec = importlib.util.spec_from_file_location("make_golden_pretrain", os.path.join(HERE, "golden", "make_golden_pretrain.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.gpu
@pytest.mark.parametrize("idx", [0, 1])
def test_pretrain_head_matches_reference(vsa, idx):
    mk = _maker()
    c = mk.CASES[idx]
    g = np.load(os.path.join(HERE, "golden", "pretrain_golden.npz"))
    m = vsa.PretrainModel(feature_dim=c["d"], num_heads=c["H"], num_layers=c["L"], dropout=0.3).eval()
    assert sorted(k for k in m.state_dict() if not k.startswith("encoder.")) == ["video_transform.bias", "video_transform.weight"]
    m.encoder.load_state_dict(vsa.synth.make_state_dict(c["d"], c["L"], c["wseed"]), strict=True)
    w, b = mk.head_weights(c["d"], c["wseed"] + 1)
    with torch.no_grad():
        m.video_transform.weight.copy_(w)
        m.video_transform.bias.copy_(b)
    m = m.to(DEV)
    x, mask, vid = (t.to(DEV) for t in mk.inputs(c))
    loss, center, repel = m(x, vid, mask, pen_met=c["pen"])
    got = np.array([loss.item(), center.item(), repel.item()])
    assert np.abs(got - g[c["name"] + "_losses"]).max() < 5e-6, (got, g[c["name"] + "_losses"])
    (loss + 0.5 * center + repel).backward()                   # pretrain.py:64
    gv = m.video_transform.weight.grad.cpu().numpy()
    assert np.abs(gv[:8] - g[c["name"] + "_grad_vt_rows"]).max() < 1e-6
    assert abs(np.linalg.norm(gv.astype(np.float64)) - g[c["name"] + "_grad_vt_norm"][0]) < 1e-6
    assert np.abs(m.encoder.final_layer.weight.grad.cpu().numpy() - g[c["name"] + "_grad_final"]).max() < 1e-6


def test_repelling_loss_equals_the_materialised_form(vsa):
    """The O(T d) identity against the reference's own formulation (simnet_pretrain.py:56-69) on random data."""
    m = vsa.PretrainModel(feature_dim=128, num_heads=4, num_layers=1)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 50, 64, generator=g)
    mask = torch.zeros(3, 50, dtype=torch.bool)
    mask[1, 30:] = True
    xm = x * (mask == False).unsqueeze(2)                      # noqa: E712
    xn = xm / (xm.norm(dim=2, keepdim=True) + 1e-9)
    sim = torch.matmul(xn, xn.transpose(1, 2)) * (torch.eye(50) == 0).float().unsqueeze(0)
    want = sim.mean(dim=1).mean()
    assert abs(m.repelling_loss(x, mask).item() - want.item()) < 1e-6
    assert abs(m.repelling_loss(x, None).item() -
               (torch.matmul(x / (x.norm(dim=2, keepdim=True) + 1e-9), (x / (x.norm(dim=2, keepdim=True) + 1e-9)).transpose(1, 2))
                * (torch.eye(50) == 0).float()).mean(dim=1).mean().item()) < 1e-6
