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


def _head_reference(hidden, logits, vid, mask, W, b, temp, pen):
    """The reference's formulas (simnet_pretrain.py:35-98) restated op for op in float64 torch, [T,T] matrix included."""
    import torch.nn.functional as F
    feats = F.linear(hidden, W, b)
    x = feats * (mask == False).unsqueeze(2) if mask is not None else feats      # noqa: E712
    x = x / (x.norm(dim=2, keepdim=True) + 1e-9)
    T = x.shape[1]
    sim = torch.matmul(x, x.transpose(1, 2)) * (torch.eye(T, dtype=x.dtype) == 0).to(x.dtype).unsqueeze(0)
    repel = sim.mean(dim=1).mean()
    sc = logits
    if mask is not None:
        sc = sc.masked_fill(mask.unsqueeze(2), float("-inf"))
    mix = F.softmax(sc / temp, dim=1)
    if pen == "entropy":
        e = (mix + 1e-9) * torch.log(mix + 1e-9)
        if mask is not None:
            e = e.masked_fill(mask.unsqueeze(2), 0.)
        center = e.mean(dim=1).mean()
    else:
        center = torch.norm(mix, dim=1).mean()
    pooled = torch.matmul(mix.transpose(1, 2), feats).squeeze(1)
    loss = (-F.softmax(vid, dim=1) * torch.log(F.softmax(pooled, dim=1))).mean()
    return loss, center, repel


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,d,pen,masked", [(3, 150, 256, "entropy", True), (2, 64, 128, "norm", True), (1, 333, 512, "entropy", False),
                                              (4, 65, 256, "norm", False)])
def test_pretrain_head_kernels_match_float64_formulas(vsa, B, T, d, pen, masked):
    """_PretrainHead (video_transform + repel + pooling + penalties + soft CE, forward and backward kernels) against
    the reference's formulas in float64, for a weighted sum of the three losses (pretrain.py:62)."""
    from importlib import import_module
    head = import_module("video-summarization_amd.pretrain")._PretrainHead
    g = torch.Generator().manual_seed(B * T + d)
    hidden = torch.randn(B, T, d, generator=g, dtype=torch.float64)
    logits = torch.randn(B, T, 1, generator=g, dtype=torch.float64)
    vid = torch.randn(B, 512, generator=g, dtype=torch.float64)
    W = torch.randn(512, d, generator=g, dtype=torch.float64) / d ** 0.5
    bias = 0.1 * torch.randn(512, generator=g, dtype=torch.float64)
    mask = None
    if masked:
        mask = torch.zeros(B, T, dtype=torch.bool)
        for i in range(B):
            mask[i, T - 7 * i - 3:] = True
    leaves = [t.clone().requires_grad_(True) for t in (hidden, logits, W, bias)]
    want = _head_reference(leaves[0], leaves[1], vid, mask, leaves[2], leaves[3], 0.4, pen)
    (want[0] + 0.5 * want[1] + 1.0 * want[2]).backward()
    dl = [t.detach().float().to(DEV).requires_grad_(True) for t in (hidden, logits, W, bias)]
    got = head.apply(dl[0], dl[1], vid.float().to(DEV), None if mask is None else mask.to(DEV), dl[2], dl[3], 0.4, pen == "entropy")
    (got[0] + 0.5 * got[1] + 1.0 * got[2]).backward()
    torch.cuda.synchronize()
    for i, name in enumerate(("distillation", "centering", "repelling")):
        assert abs(got[i].item() - want[i].item()) < 2e-6 * max(1.0, abs(want[i].item())), (name, got[i].item(), want[i].item())
    for a, r, name in zip(dl, leaves, ("d_hidden", "d_logits", "d_weight", "d_bias")):
        err = (a.grad.double().cpu() - r.grad).abs().max().item()
        scale = r.grad.abs().max().item()
        assert err <= 2e-5 * scale + 1e-9, "%s: err %.3e, max %.3e" % (name, err, scale)


@pytest.mark.gpu
def test_pretrain_loop_like_the_reference(vsa):
    """pretrain.py:49-86 in shape: autocast, the three losses combined (main + 0.5 center + repel), GradScaler, Adam on
    the ENCODER's parameters only (:40), sentinel-padded batches with the caller-side mask (:57).  The loss must fall
    and two runs from one torch seed must be bit-identical (HIP encoder with dropout + HIP head, ordered reductions)."""
    synth = vsa.synth

    def run():
        torch.manual_seed(4321)
        m = vsa.PretrainModel(num_heads=4, feature_dim=256, num_layers=2, sparsity=0.5, dropout=0.2, num_classes=1,
                              use_pos=True).to(DEV)
        m.encoder.load_state_dict(synth.make_state_dict(256, 2, 3))
        opt = torch.optim.Adam(m.encoder.parameters(), lr=1e-4, weight_decay=5e-4)
        scaler = torch.amp.GradScaler("cuda")
        features = synth.make_features(4, 90, 8, "pool5", [90, 71, 60, 33]).to(DEV)
        vid_rep = torch.randn(4, 512, generator=torch.Generator().manual_seed(2)).to(DEV)
        m.train()
        losses = []
        for _ in range(10):
            mask = (features[:, :, 0] == 1000)                                   # pretrain.py:57
            with torch.amp.autocast("cuda"):
                main_loss, center_loss, repel_loss = m(features, vid_rep, mask)
                loss = main_loss + center_loss * 0.5 + 1. * repel_loss           # pretrain.py:62
            opt.zero_grad()
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            losses.append(loss.item())
        assert m.video_transform.weight.grad is not None                        # the head's own Linear gets gradients too
        return losses, [p.detach().clone() for p in m.parameters()]

    l1, p1 = run()
    l2, p2 = run()
    assert all(np.isfinite(l1)) and min(l1[-3:]) < l1[0]
    assert l1 == l2 and all(torch.equal(a, b) for a, b in zip(p1, p2))
