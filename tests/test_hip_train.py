"""GPU tests of the HIP TRAINING path (C ABI include/vs_train.h), through the C ABI.

Checkers: (1) gradients produced by the IMPORTED reference in float64 (tests/golden/make_golden_train.py: reference
SimNet in train mode with dropout 0 + reference utils.mse_with_mask_loss); (2) a float64 torch restatement with
EXPLICIT dropout masks (tests/torch_ref.py) fed with the very masks the library draws, for everything dropout touches.

Tolerances (fp32 arithmetic against a float64 truth): per tensor max|err| <= 1e-4 absolute AND <= 1e-3 of the tensor's
largest entry (+1e-6 for gradients that are analytically zero, e.g. the key bias: a sum that cancels).  Observed: ~1e-6 relative."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
import torch_ref
import tolerances as tol

pytestmark = pytest.mark.gpu
ATOL, RTOL = tol.TRAIN_GRAD_ATOL, tol.TRAIN_GRAD_RTOL


def _dev():
    assert torch.cuda.is_available(), "these tests need a HIP device"
    return torch.device("cuda:0")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _close(got, want, what="", atol=ATOL, rtol=RTOL):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    err = (got - want).abs().max().item()
    scale = want.abs().max().item()
    assert (atol is None or err <= atol) and err <= rtol * scale + 1e-6, "%s: max err %.3e (max |want| %.3e)" % (what, err, scale)
    return err / (scale + 1e-30)


def train_cases():
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


# ---------------------------------------------------------------------------------------------
# whole model: gradients against the imported reference (float64)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", train_cases(), ids=lambda c: c["name"])
def test_gradients_match_reference_golden(vsa, case):
    c = case
    z = np.load(os.path.join(GOLDEN, c["name"] + ".npz"))
    sd = vsa.synth.make_state_dict(c["d"], c["L"], c["wseed"])
    x, mask, target, R = _inputs(vsa.synth, c)
    m = vsa.SimNet(num_heads=c["H"], d_model=c["d"], num_layers=c["L"], sparsity=0.0, dropout=0.0)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).train()
    xd = x.to(_dev()).requires_grad_(True)
    md = None if mask is None else mask.to(_dev())
    pred, hidden = m(xd, md)
    assert pred.requires_grad and hidden.requires_grad
    mk = md if md is not None else torch.zeros(x.shape[:2], dtype=torch.bool, device=_dev())
    loss = vsa.mse_with_mask_loss(pred, target.to(_dev()), mk)                       # train.py:122
    if c["hidden_w"]:
        loss = loss + c["hidden_w"] * (hidden * R.to(_dev())).sum()
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(z["loss"])) <= 2e-6 * max(1.0, abs(float(z["loss"])))
    valid = torch.ones(x.shape[:2], dtype=torch.bool) if mask is None else ~mask
    assert (pred.detach().cpu() - torch.from_numpy(z["logits"]))[valid].abs().max().item() < 1e-4
    grads = {"x": xd.grad}
    grads.update({k: p.grad for k, p in m.named_parameters()})
    keys = json.loads(str(z["keys"]))
    assert sorted(keys) == sorted(grads.keys())
    worst = 0.0
    for k in keys:
        g = grads[k]
        assert g is not None and torch.isfinite(g).all(), k
        g2 = g.reshape(-1, g.shape[-1]) if g.dim() > 1 else g.reshape(1, -1)
        rows = torch.from_numpy(z["r:" + k])
        want = torch.from_numpy(z["g:" + k])
        tot, nrm, gmax, ref32 = z["s:" + k]
        got = g2[rows.to(g2.device)].double().cpu()
        err = (got - want.double()).abs().max().item()
        assert err <= ATOL and err <= RTOL * gmax + 1e-6, "%s: err %.3e, max|g| %.3e (reference fp32 own err %.3e)" % (k, err, gmax, ref32)
        # whole-tensor checks for the sampled ones: sum and L2 norm
        # whole-tensor sum: the per-element bound (RTOL * gmax) allows a random-walk sum error of sqrt(numel) times that; a
        # quarter of it is granted on top of the relative term (the 768 x 768 key-projection gradient sums to ~0: every
        # row of it is orthogonal to the softmax's shift invariance, so the sum is pure rounding)
        assert abs(g.double().sum().item() - tot) <= 1e-3 * max(abs(tot), nrm) + 0.25 * RTOL * gmax * g.numel() ** 0.5 + 1e-7, k
        assert abs(g.double().norm().item() - nrm) <= 1e-4 * nrm + 1e-7, k
        worst = max(worst, err / (gmax + 1e-12) if gmax > 1e-6 else 0.0)
    print("%s: worst gradient error relative to the tensor's max: %.2e" % (c["name"], worst))


def test_train_mode_without_dropout_equals_the_scoring_path(vsa):
    """Dropout 0: the training forward and the scoring kernels compute the same function (different kernels)."""
    sd = vsa.synth.make_state_dict(256, 2, 3)
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=2, sparsity=0.0, dropout=0.0)
    m.load_state_dict(sd)
    m = m.to(_dev())
    x = vsa.synth.make_features(2, 150, 4, "randn", [150, 99]).to(_dev())
    mask = vsa.synth.padding_mask(x)
    a, ha = m.train()(x, mask)
    with torch.no_grad():
        b, hb = m.eval()(x, mask)
    valid = ~mask
    assert a.requires_grad and not b.requires_grad
    assert (a.detach() - b)[valid].abs().max().item() < 2e-5 and (ha.detach() - hb)[valid].abs().max().item() < 5e-5


def test_embedding_dropout_does_not_exist_without_a_positional_table(vsa):
    """the reference's embedding dropout lives INSIDE PositionalEncoding (simnet.py:224,237), so a
    ``use_pos=False`` model has none.  The C ABI (``vs_train_forward`` / ``_backward``) ignores ``p_embed`` for a
    handle packed without a positional table; with one, the same ``p_embed`` does change the result."""
    lib = vsa._lib.load()
    x = vsa.synth.make_features(2, 70, 4, "randn").to(_dev())
    for use_pos in (False, True):
        sd = vsa.synth.make_state_dict(256, 1, 5, use_pos=use_pos)
        m = vsa.SimNet(num_heads=4, d_model=256, num_layers=1, sparsity=0.5, dropout=0.0, use_pos=use_pos)
        m.load_state_dict(sd)
        m = m.to(_dev())
        packed = m._packed_weights(_dev())
        outs = []
        for p_embed in (0.0, 0.5):
            cfg = vsa._lib.DropoutCfg(p_embed, 0.0, 77)
            scores = torch.empty((2, 70, 1), device=_dev())
            saved = torch.empty((lib.vs_train_saved_bytes(packed.handle, 2, 70),), dtype=torch.uint8, device=_dev())
            ws = torch.empty((lib.vs_train_workspace_bytes(packed.handle, 2, 70),), dtype=torch.uint8, device=_dev())
            vsa._lib.check(lib.vs_train_forward(packed.handle, x.data_ptr(), None, 2, 70, C.byref(cfg), scores.data_ptr(), None,
                                                saved.data_ptr(), saved.numel(), ws.data_ptr(), ws.numel(), _stream()))
            torch.cuda.synchronize()
            outs.append(scores.clone())
        if use_pos:
            assert not torch.equal(outs[0], outs[1])
        else:
            assert torch.equal(outs[0], outs[1])
        # and the module's own train-mode forward (sparsity 0.5) equals eval for the use_pos=False model
        if not use_pos:
            a, _ = m.train()(x)
            with torch.no_grad():
                b, _ = m.eval()(x)
            assert (a.detach() - b).abs().max().item() < 2e-5


def test_parameter_updates_reach_every_kernel_layout_copy(vsa):
    """Since round 3 the kernel-layout weight images (fragment-major fp32 / fp16x3, bf16 LDS images, dgrad transposes)
    are rebuilt LAZILY, per family, by the first call that reads them after a ``vs_weights_update``: every compute mode
    and the backward must see an in-place parameter write (optimizer step) exactly like a freshly packed module does."""
    sd = vsa.synth.make_state_dict(256, 2, 9)
    x_small = vsa.synth.make_features(1, 90, 3, "randn").to(_dev())          # latency kernels (fragment-major copies)
    x_big = vsa.synth.make_features(20, 1024, 4, "randn").to(_dev())        # 20480 rows: LDS-tiled kernels

    def fresh(state):
        mm = vsa.SimNet(num_heads=4, d_model=256, num_layers=2, sparsity=0.0, dropout=0.0)
        mm.load_state_dict(state)
        return mm.to(_dev()).eval()

    m = fresh(sd)
    with torch.no_grad():
        for mode in ("fp32", "fp16x3", "bf16"):                                 # build every family once
            m.set_compute_dtype(mode)
            m(x_small); m(x_big)
        for name, prm in m.named_parameters():                                  # an "optimizer step": in-place writes
            prm.add_(0.01 * torch.randn_like(prm))
        sd2 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        ref = fresh(sd2)
        for mode in ("fp32", "fp16x3", "bf16"):
            m.set_compute_dtype(mode); ref.set_compute_dtype(mode)
            for xx in (x_small, x_big):
                a, ha = m(xx)
                b, hb = ref(xx)
                assert torch.equal(a, b) and torch.equal(ha, hb), mode
    # backward after an update: gradients equal a fresh module's
    m.set_compute_dtype("fp32"); ref.set_compute_dtype("fp32")
    grads = []
    for mod in (m, ref):
        mod.train()
        mod.zero_grad(set_to_none=True)
        p1, _ = mod(x_small)
        p1.square().mean().backward()
        grads.append([p.grad.clone() for p in mod.parameters()])
    for ga, gb in zip(*grads):
        assert torch.equal(ga, gb)


def test_training_has_no_cpu_fallback(vsa):
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=1).train()
    with pytest.raises(RuntimeError, match="HIP"):
        m(torch.zeros(1, 8, 1024))


# ---------------------------------------------------------------------------------------------
# kernels in isolation
# ---------------------------------------------------------------------------------------------
def _qkv(B, H, T, dh, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(B, H, T, dh, generator=g, dtype=torch.float64) for _ in range(3)]


@pytest.mark.parametrize("B,H,T,dh,masked,p", [(2, 4, 320, 64, False, 0.0), (1, 2, 777, 64, False, 0.0), (2, 4, 200, 64, True, 0.0),
                                                (2, 8, 65, 32, True, 0.0), (1, 4, 150, 128, True, 0.0), (1, 2, 33, 128, False, 0.0),
                                                (2, 4, 130, 64, True, 0.3), (1, 8, 97, 32, False, 0.5), (1, 2, 260, 128, True, 0.2)])
def test_attention_forward_and_backward_kernels(vsa, B, H, T, dh, masked, p):
    """Forward (values + saved log-sum-exp) and backward (dq | dk | dv) against float64 torch autograd, with the
    library's own dropout mask applied explicitly in the checker."""
    lib = vsa._lib.load()
    q, k, v = _qkv(B, H, T, dh, 100 + T)
    q.requires_grad_(True), k.requires_grad_(True), v.requires_grad_(True)
    mask = vsa.synth.random_mask(B, T, 5) if masked else None
    scale = (H * dh) ** -0.5
    seed, site = 0x1234567887654321, 7
    keep = None
    if p > 0:
        kd = torch.empty(B, H, T, T, dtype=torch.uint8, device=_dev())
        vsa._lib.check(lib.vs_train_dropout_mask_attention(kd.data_ptr(), B, H, T, seed, site, p, _stream()))
        keep = kd.cpu()
        rate = keep.double().mean().item()
        assert abs(rate - (1 - p)) < 4 * math.sqrt(p * (1 - p) / keep.numel()) + 1e-3, rate
    want, lse2 = torch_ref.attention_with_mask(q, k, v, mask, scale, keep, p)
    dO = torch.randn(B, T, H * dh, generator=torch.Generator().manual_seed(9), dtype=torch.float64)
    want.backward(dO)
    qd, kd_, vd = (t.detach().float().to(_dev()).contiguous() for t in (q, k, v))
    md = None if mask is None else mask.to(_dev()).view(torch.uint8)
    out = torch.empty(B, T, H * dh, device=_dev())
    lse = torch.empty(B, H, T, device=_dev())
    vsa._lib.check(lib.vs_train_attention_forward(qd.data_ptr(), kd_.data_ptr(), vd.data_ptr(), None if md is None else md.data_ptr(),
                                                  out.data_ptr(), lse.data_ptr(), B, H, T, dh, scale, seed, site, p, _stream()))
    _close(out, want, "attention out")
    assert (lse.cpu().double() - lse2.detach()).abs().max().item() < 1e-4
    dqkv = torch.empty(B, T, 3 * H * dh, device=_dev())
    scratch = torch.empty(B * H * T, device=_dev())
    dOd = dO.float().to(_dev())
    vsa._lib.check(lib.vs_train_attention_backward(qd.data_ptr(), kd_.data_ptr(), vd.data_ptr(), None if md is None else md.data_ptr(),
                                                   out.data_ptr(), dOd.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), scratch.data_ptr(),
                                                   B, H, T, dh, scale, seed, site, p, _stream()))
    torch.cuda.synchronize()
    d = H * dh
    tok = lambda g: g.permute(0, 2, 1, 3).reshape(B, T, d)       # noqa: E731  head-major grad -> token-major
    _close(dqkv[:, :, :d], tok(q.grad), "dq")
    _close(dqkv[:, :, d:2 * d], tok(k.grad), "dk")
    _close(dqkv[:, :, 2 * d:], tok(v.grad), "dv")


@pytest.mark.parametrize("B,H,T,dh,masked,p", [(2, 4, 320, 64, False, 0.0), (1, 2, 777, 64, False, 0.0), (2, 4, 200, 64, True, 0.0),
                                                (2, 8, 65, 32, True, 0.0), (1, 4, 1, 64, False, 0.0), (2, 4, 130, 64, True, 0.3),
                                                (1, 8, 97, 32, False, 0.5), (1, 4, 513, 64, True, 0.2),
                                                (2, 4, 200, 128, True, 0.0), (1, 2, 131, 128, False, 0.3), (1, 2, 64, 128, True, 0.5)])
def test_attention_forward_and_backward_kernels_bf16(vsa, B, H, T, dh, masked, p):
    """The training attention on the bf16 matrix pipe (VS_TRAIN_FLAG_BF16_ATTENTION): forward (values + log-sum-exp) and
    backward (dq | dk | dv) against float64 torch autograd on UNROUNDED operands, the library's own dropout mask applied
    explicitly in the checker.  Bounds relative to each tensor's largest entry: the bf16 rounding of q, k, v, dO, P and dS
    (2^-9 each) - a wrong index map of the row / transposed LDS fragments would be an O(1) error.  Run twice: bitwise
    reproducible."""
    lib = vsa._lib.load()
    q, k, v = _qkv(B, H, T, dh, 100 + T)
    q.requires_grad_(True), k.requires_grad_(True), v.requires_grad_(True)
    mask = vsa.synth.random_mask(B, T, 5) if masked else None
    scale = (H * dh) ** -0.5
    seed, site = 0x1234567887654321, 7
    keep, dbits = None, None
    if p > 0:
        kd = torch.empty(B, H, T, T, dtype=torch.uint8, device=_dev())
        vsa._lib.check(lib.vs_train_dropout_mask_attention(kd.data_ptr(), B, H, T, seed, site, p, _stream()))
        keep = kd.cpu()
        dbits = torch.empty(lib.vs_train_attention_dropout_bits_bytes(B, H, T), dtype=torch.uint8, device=_dev())
        vsa._lib.check(lib.vs_train_attention_dropout_bits(dbits.data_ptr(), B, H, T, seed, site, p, _stream()))
        # the bit-packed copies hold exactly the decisions of the byte dump, in both orientations
        W = (T + 31) // 32
        words = dbits.view(torch.int32).cpu().view(2, B * H, T, W)
        bit = lambda wds, j: (wds[..., j // 32] >> (j % 32)) & 1      # noqa: E731
        kq = torch.stack([bit(words[0], j) for j in range(T)], dim=-1).view(B, H, T, T)        # [.., query, key]
        kk = torch.stack([bit(words[1], j) for j in range(T)], dim=-1).view(B, H, T, T)        # [.., key, query]
        assert torch.equal(kq.to(torch.uint8), keep) and torch.equal(kk.transpose(2, 3).to(torch.uint8), keep)
    want, lse2 = torch_ref.attention_with_mask(q, k, v, mask, scale, keep, p)
    dO = torch.randn(B, T, H * dh, generator=torch.Generator().manual_seed(9), dtype=torch.float64)
    want.backward(dO)
    qd, kd_, vd = (t.detach().float().to(_dev()).contiguous() for t in (q, k, v))
    md = None if mask is None else mask.to(_dev()).view(torch.uint8)
    dOd = dO.float().to(_dev())
    d = H * dh
    runs = []
    for _ in range(2):
        out = torch.full((B, T, d), float("nan"), device=_dev())
        lse = torch.full((B, H, T), float("nan"), device=_dev())
        vsa._lib.check(lib.vs_train_attention_forward_bf16(qd.data_ptr(), kd_.data_ptr(), vd.data_ptr(), None if md is None else md.data_ptr(),
                                                           out.data_ptr(), lse.data_ptr(), B, H, T, dh, scale, p,
                                                           None if dbits is None else dbits.data_ptr(), 0, _stream()))
        dqkv = torch.full((B, T, 3 * d), float("nan"), device=_dev())
        scratch = torch.empty(B * H * T, device=_dev())
        vsa._lib.check(lib.vs_train_attention_backward_bf16(qd.data_ptr(), kd_.data_ptr(), vd.data_ptr(), None if md is None else md.data_ptr(),
                                                            out.data_ptr(), dOd.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), scratch.data_ptr(),
                                                            B, H, T, dh, scale, p, None if dbits is None else dbits.data_ptr(), 0, _stream()))
        torch.cuda.synchronize()
        runs.append((out.clone(), lse.clone(), dqkv.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*runs))
    # the bf16-STORED form (what the training forward saves when both low-precision flags are set: q times scale * log2 e,
    # k, v as bf16 planes) gives the same bits: the kernels round the fp32-stored values to exactly these
    q16 = (qd * (scale * 1.4426950408889634)).to(torch.bfloat16).contiguous()
    k16, v16 = kd_.to(torch.bfloat16).contiguous(), vd.to(torch.bfloat16).contiguous()
    out16 = torch.full((B, T, d), float("nan"), device=_dev())
    lse16 = torch.full((B, H, T), float("nan"), device=_dev())
    vsa._lib.check(lib.vs_train_attention_forward_bf16(q16.data_ptr(), k16.data_ptr(), v16.data_ptr(), None if md is None else md.data_ptr(),
                                                       out16.data_ptr(), lse16.data_ptr(), B, H, T, dh, scale, p,
                                                       None if dbits is None else dbits.data_ptr(), 1, _stream()))
    dqkv16 = torch.full((B, T, 3 * d), float("nan"), device=_dev())
    vsa._lib.check(lib.vs_train_attention_backward_bf16(q16.data_ptr(), k16.data_ptr(), v16.data_ptr(), None if md is None else md.data_ptr(),
                                                        out16.data_ptr(), dOd.data_ptr(), lse16.data_ptr(), dqkv16.data_ptr(), scratch.data_ptr(),
                                                        B, H, T, dh, scale, p, None if dbits is None else dbits.data_ptr(), 1, _stream()))
    torch.cuda.synchronize()
    assert torch.equal(out16, runs[0][0]) and torch.equal(lse16, runs[0][1]) and torch.equal(dqkv16, runs[0][2])
    out, lse, dqkv = runs[0]
    assert torch.isfinite(out).all() and torch.isfinite(dqkv).all()
    _close(out, want, "attention out (bf16)", atol=None, rtol=1.5e-2)
    assert (lse.cpu().double() - lse2.detach()).abs().max().item() < 2e-2
    tok = lambda g: g.permute(0, 2, 1, 3).reshape(B, T, d)       # noqa: E731  head-major grad -> token-major
    # a one-key row has dS = P (dP - delta) = 0 analytically; in bf16 dP (rounded operands) and delta (fp32 row dot) no
    # longer cancel to the last bit, so dq / dk get an absolute floor of bf16 rounding of an O(1) dP times the scale
    floor = 3e-3 if T == 1 else 0.0
    for name, got, ref in (("dq", dqkv[:, :, :d], tok(q.grad)), ("dk", dqkv[:, :, d:2 * d], tok(k.grad))):
        if floor and ref.abs().max().item() < floor:
            assert got.abs().max().item() < floor, name
        else:
            _close(got, ref, name + " (bf16)", atol=None, rtol=3e-2)
    _close(dqkv[:, :, 2 * d:], tok(v.grad), "dv (bf16)", atol=None, rtol=3e-2)


@pytest.mark.parametrize("p", [0.0, 0.3])
def test_bf16_training_storage_forms_agree_bit_for_bit(vsa, lp_train_everywhere, p):
    """The bf16 training mode STORES the tensors that are only ever bf16 matrix operands as bf16 (q times scale * log2 e, k, v;
    the MLP hidden tensor; the gated gradient of that tensor) - their producers' epilogues round them, their consumers read
    them as they are.  VS_LP_STORE32 = 1 keeps them fp32, rounded by every consumer on its way into LDS: the same bf16 values,
    so logits, hidden states and every gradient agree BIT for bit - except the bias gradients that are column sums of a
    bf16-stored gradient (fc1.bias: the gated hidden gradient; q / k / v bias: dq | dk | dv of the attention backward),
    which sum the stored (rounded) values in one form and the fp32 values in the other."""
    res = {}
    try:
        for store32 in (1, 0):
            vsa._lib.set_option("VS_LP_STORE32", store32)
            m = vsa.SimNet(num_heads=4, d_model=256, num_layers=2, sparsity=0.0, dropout=p)
            m.load_state_dict(vsa.synth.make_state_dict(256, 2, 3))
            m = m.to(_dev()).train().set_train_dtype("bf16")
            x = torch.randn(2, 200, 1024, generator=torch.Generator().manual_seed(1)).to(_dev()).requires_grad_(True)
            torch.manual_seed(5)
            pred, hid = m(x, None)
            ((pred ** 2).mean() + 1e-3 * hid.sum()).backward()
            res[store32] = [("pred", pred.detach().clone()), ("hidden", hid.detach().clone()), ("dx", x.grad.clone())] + \
                           [(n, q.grad.clone()) for n, q in m.named_parameters()]
    finally:
        vsa._lib.set_option("VS_LP_STORE32", -1)
    for (n, a), (_n, b) in zip(res[1], res[0]):
        if n.endswith("mlp.fc1.bias") or n.endswith(".sa.q.bias") or n.endswith(".sa.v.bias"):
            assert (a - b).abs().max().item() <= 2e-3 * a.abs().max().item(), n
        elif n.endswith(".sa.k.bias"):          # analytically zero
            assert max(a.abs().max().item(), b.abs().max().item()) <= tol.TRAIN_LP_ZERO_ATOL, n
        else:
            assert torch.equal(a, b), n


def test_backward_reads_the_record_in_the_form_the_forward_wrote(vsa, lp_train_everywhere):
    """the forward publishes the form of its activation record (vs_train_last_format) and the backward
    takes it back through vs_dropout_cfg.reserved.  Flipping the storage / kernel-choice switches BETWEEN a forward and its
    backward (another model, a retained graph, a test harness) therefore changes nothing: every gradient is bit-identical to
    the undisturbed step.  Before, the backward derived the form again and read bf16 planes as fp32."""
    def step(flip):
        m = vsa.SimNet(num_heads=4, d_model=256, num_layers=2, sparsity=0.0, dropout=0.0)
        m.load_state_dict(vsa.synth.make_state_dict(256, 2, 3))
        m = m.to(_dev()).train().set_train_dtype("bf16")
        x = torch.randn(2, 200, 1024, generator=torch.Generator().manual_seed(1)).to(_dev()).requires_grad_(True)
        pred, hid = m(x, None)
        loss = (pred ** 2).mean() + 1e-3 * hid.sum()
        try:
            if flip:
                vsa._lib.set_option("VS_LP_STORE32", 1)
                vsa._lib.set_option("VS_LP_MLP_UNFUSED", 1)
            loss.backward()
            torch.cuda.synchronize()
        finally:
            vsa._lib.set_option("VS_LP_STORE32", -1)
            vsa._lib.set_option("VS_LP_MLP_UNFUSED", -1)
        assert m.last_train_dtype == "bf16"
        return [("dx", x.grad.clone())] + [(n, q.grad.clone()) for n, q in m.named_parameters()]
    for (n, a), (_n, b) in zip(step(False), step(True)):
        assert torch.isfinite(a).all() and torch.equal(a, b), n


def test_low_precision_request_below_the_threshold_says_so(vsa):
    """below VS_TRAIN_LP_MIN_ROWS frames per batch the exact kernels run whatever set_train_dtype asked for -
    the model now records what ran (last_train_dtype) and warns once instead of doing so silently."""
    import warnings
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=1, sparsity=0.0, dropout=0.0)
    m.load_state_dict(vsa.synth.make_state_dict(256, 1, 3))
    m = m.to(_dev()).train().set_train_dtype("bf16")
    x = torch.randn(2, 100, 1024, generator=torch.Generator().manual_seed(1)).to(_dev())
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        m(x, None)[0].sum().backward()
        m(x, None)[0].sum().backward()
    assert m.last_train_dtype == "fp32"
    assert sum("exact fp32 kernels" in str(w.message) for w in rec) == 1
    try:
        vsa._lib.set_option("VS_TRAIN_LP_MIN_ROWS", 0)
        m(x, None)[0].sum().backward()
        assert m.last_train_dtype == "bf16"
    finally:
        vsa._lib.set_option("VS_TRAIN_LP_MIN_ROWS", -1)


@pytest.mark.parametrize("p,B,T,d", [(0.0, 2, 200, 256), (0.3, 3, 171, 256), (0.5, 1, 1, 256), (0.3, 2, 150, 512), (0.0, 1, 333, 512)])
def test_a_stationary_mlp_gemms_equal_the_tiled_ones_bit_for_bit(vsa, lp_train_everywhere, p, B, T, d):
    """d_model 256 / 512 in the bf16 training mode: fc1 (+ ReLU + dropout) and the fc2 input gradient (+ gate) run A-stationary
    (vs_train_gemm_rows.hip: a wave keeps its 32 rows of A in registers, W streams past, the bf16 result is written once).
    Same operands, same rounding points, bias first and k ascending like gemm_nt_128: VS_LP_MLP_UNFUSED = 1 (the tiled
    kernels) gives the same bits everywhere - logits, hidden states, every gradient; ragged row counts included."""
    res = {}
    try:
        for tiled in (1, 0):
            vsa._lib.set_option("VS_LP_MLP_UNFUSED", 1 if tiled else 2)       # 2: A-stationary whatever the batch size
            m = vsa.SimNet(num_heads=4, d_model=d, num_layers=2, sparsity=0.0, dropout=p)
            m.load_state_dict(vsa.synth.make_state_dict(d, 2, 3))
            m = m.to(_dev()).train().set_train_dtype("bf16")
            x = torch.randn(B, T, 1024, generator=torch.Generator().manual_seed(1)).to(_dev()).requires_grad_(True)
            torch.manual_seed(5)
            pred, hid = m(x, None)
            ((pred ** 2).mean() + 1e-3 * hid.sum()).backward()
            res[tiled] = [("pred", pred.detach().clone()), ("hidden", hid.detach().clone()), ("dx", x.grad.clone())] + \
                         [(n, q.grad.clone()) for n, q in m.named_parameters()]
    finally:
        vsa._lib.set_option("VS_LP_MLP_UNFUSED", -1)
    for (n, a), (_n, b) in zip(res[1], res[0]):
        assert torch.equal(a, b), n


@pytest.mark.parametrize("M,N,K", [(300, 256, 1024), (4096, 1024, 256), (77, 768, 256), (1000, 64, 192), (5000, 512, 2048),
                                   (16, 256, 256), (1, 128, 64)])
def test_wgrad_kernel(vsa, M, N, K):
    lib = vsa._lib.load()
    g = torch.Generator().manual_seed(M + N)
    dY = torch.randn(M, N, generator=g, dtype=torch.float64)
    X = torch.randn(M, K, generator=g, dtype=torch.float64)
    dYd, Xd = dY.float().to(_dev()), X.float().to(_dev())
    dW = torch.full((N, K), float("nan"), device=_dev())
    db = torch.full((N,), float("nan"), device=_dev())
    scratch = torch.empty(lib.vs_train_wgrad_scratch_floats(M, N, K), device=_dev())
    for _ in range(2):       # twice: results must be bit-identical (fixed reduction order)
        vsa._lib.check(lib.vs_train_wgrad(dYd.data_ptr(), Xd.data_ptr(), M, N, K, dW.data_ptr(), db.data_ptr(),
                                          scratch.data_ptr(), _stream()))
        torch.cuda.synchronize()
        first = (dW.clone(), db.clone()) if _ == 0 else first
    assert torch.equal(first[0], dW) and torch.equal(first[1], db)
    # unnormalised N(0,1) operands: entries grow like sqrt(M), so the bound is relative (fp32: ~1e-6 observed)
    _close(dW, dY.t() @ X, "dW", atol=None, rtol=2e-5)
    _close(db, dY.sum(0), "db", atol=None, rtol=2e-5)


@pytest.mark.parametrize("M,N,K", [(4096, 256, 1024), (1000, 768, 256), (333, 128, 132), (65, 4, 8), (8192, 1024, 256)])
def test_wgrad_bf16_kernel(vsa, M, N, K):
    """The weight-gradient GEMM on the bf16 matrix pipe (low-precision training): against float64 on the SAME bf16-rounded
    operands (every index map of the transposed LDS reads is then pinned to fp32 rounding: 2e-5 relative), the bias
    gradient against the unrounded column sums, and twice for bitwise reproducibility."""
    lib = vsa._lib.load()
    g = torch.Generator().manual_seed(M + N + 1)
    dY = torch.randn(M, N, generator=g)
    X = torch.randn(M, K, generator=g)
    dYd, Xd = dY.to(_dev()), X.to(_dev())
    dW = torch.full((N, K), float("nan"), device=_dev())
    db = torch.full((N,), float("nan"), device=_dev())
    scratch = torch.empty(lib.vs_train_wgrad_scratch_floats(M, N, K), device=_dev())
    for it in range(2):
        vsa._lib.check(lib.vs_train_wgrad_bf16(dYd.data_ptr(), Xd.data_ptr(), M, N, K, dW.data_ptr(), db.data_ptr(),
                                               scratch.data_ptr(), _stream()))
        torch.cuda.synchronize()
        first = (dW.clone(), db.clone()) if it == 0 else first
    assert torch.equal(first[0], dW) and torch.equal(first[1], db)
    r = lambda t: t.to(torch.bfloat16).double()          # noqa: E731  (round to nearest even, like v_cvt_pk_bf16_f32)
    _close(dW, r(dY).t() @ r(X), "dW", atol=None, rtol=2e-5)
    _close(db, dY.double().sum(0), "db", atol=None, rtol=2e-5)
    # and the distance to the unrounded product is the bf16 rounding itself (~2^-9 per operand, averaged over M terms)
    _close(dW, dY.double().t() @ X.double(), "dW vs unrounded", atol=None, rtol=1e-2)


@pytest.fixture
def lp_train_everywhere(vsa):
    """low-precision GEMMs from the first row on (default: above 1024 frames per batch)"""
    vsa._lib.set_option("VS_TRAIN_LP_MIN_ROWS", 0)
    yield
    vsa._lib.set_option("VS_TRAIN_LP_MIN_ROWS", -1)


@pytest.mark.parametrize("case", train_cases(), ids=lambda c: c["name"])
def test_bf16_training_gradients_within_the_low_precision_tolerance(vsa, lp_train_everywhere, case):
    """``set_train_dtype("bf16")`` - the counterpart of the reference's fp16 autocast (train.py:120): every Linear, dgrad
    and wgrad GEMM on the bf16 matrix pipe.  Loss and every gradient against the float64 goldens of the IMPORTED reference
    at tests/tolerances.py's TRAIN_LP_* (per tensor, relative to its largest entry)."""
    c = case
    z = np.load(os.path.join(GOLDEN, c["name"] + ".npz"))
    sd = vsa.synth.make_state_dict(c["d"], c["L"], c["wseed"])
    x, mask, target, R = _inputs(vsa.synth, c)
    m = vsa.SimNet(num_heads=c["H"], d_model=c["d"], num_layers=c["L"], sparsity=0.0, dropout=0.0)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).train().set_train_dtype("bf16")
    xd = x.to(_dev()).requires_grad_(True)
    md = None if mask is None else mask.to(_dev())
    pred, hidden = m(xd, md)
    mk = md if md is not None else torch.zeros(x.shape[:2], dtype=torch.bool, device=_dev())
    loss = vsa.mse_with_mask_loss(pred, target.to(_dev()), mk)
    if c["hidden_w"]:
        loss = loss + c["hidden_w"] * (hidden * R.to(_dev())).sum()
    loss.backward()
    torch.cuda.synchronize()
    want = float(z["loss"])
    assert abs(loss.item() - want) <= tol.TRAIN_LP_LOSS_RTOL * max(1.0, abs(want)), (loss.item(), want)
    grads = {"x": xd.grad}
    grads.update({k: p.grad for k, p in m.named_parameters()})
    worst, worst_k, worst_l2 = 0.0, None, 0.0
    for k in json.loads(str(z["keys"])):
        g = grads[k]
        assert g is not None and torch.isfinite(g).all(), k
        g2 = g.reshape(-1, g.shape[-1]) if g.dim() > 1 else g.reshape(1, -1)
        rows = torch.from_numpy(z["r:" + k])
        want_g = torch.from_numpy(z["g:" + k]).double()
        tot, nrm, gmax, ref32 = z["s:" + k]
        diff = g2[rows.to(g2.device)].double().cpu() - want_g
        err = diff.abs().max().item()
        l2 = diff.norm().item() / (want_g.norm().item() + 1e-30)
        # two bounds per tensor: the largest element error relative to the tensor's largest entry (a ReLU unit whose
        # pre-activation is within bf16 rounding of zero flips and moves ONE row of d_fc1 / one entry of its bias: a few
        # per cent of the maximum in these 60..800-frame batches), and the relative L2 error over the sampled rows
        if gmax < 1e-6:
            # analytically zero (k.bias: softmax is shift invariant, so the dk rows sum to zero) - with the attention
            # backward itself on bf16 operands the rows no longer cancel to fp32 rounding
            assert g.double().norm().item() <= tol.TRAIN_LP_ZERO_ATOL, "%s: |g| %.3e" % (k, g.double().norm().item())
            continue
        assert err <= tol.TRAIN_LP_GRAD_RTOL * gmax + 1e-6, "%s: err %.3e, max|g| %.3e" % (k, err, gmax)
        if k.endswith("mlp.fc1.weight") and diff.dim() > 1 and diff.shape[0] >= 4 and l2 > tol.TRAIN_LP_GRAD_L2:
            # ReLU-flip allowance (tests/tolerances.py), fc1.weight only: ONE sampled row - the fc1 unit that flipped - is set
            # aside from the L2 figure (it stays under the largest-element bound above); everything else must meet the L2 bound
            sq = diff.pow(2).sum(-1)
            keep_rows = torch.ones_like(sq, dtype=torch.bool)
            keep_rows[sq.argmax()] = False
            l2 = sq[keep_rows].sum().sqrt().item() / (want_g[keep_rows].norm().item() + 1e-30)
            assert l2 <= tol.TRAIN_LP_FC1_L2, "%s: relative L2 error %.3e with the flipped unit's row set aside" % (k, l2)
            l2 = 0.0        # (not counted into the worst figure below)
        assert l2 <= (tol.TRAIN_LP_FC1_L2 if k.endswith("mlp.fc1.bias") else tol.TRAIN_LP_GRAD_L2), "%s: relative L2 error %.3e" % (k, l2)
        assert abs(g.double().norm().item() - nrm) <= 2e-2 * nrm + 1e-7, k
        if gmax > 1e-6 and err / gmax > worst:
            worst, worst_k = err / gmax, k
        if gmax > 1e-6 and l2 > worst_l2:
            worst_l2 = l2
    assert worst > 1e-5, "the low-precision path did not run (gradients at exact-fp32 accuracy)"
    print("%s: bf16 training, worst gradient error relative to the tensor's max: %.2e (%s); worst relative L2 error %.2e" % (c["name"], worst, worst_k, worst_l2))


@pytest.mark.parametrize("case", train_cases(), ids=lambda c: c["name"])
def test_fp16_training_gradients_under_a_loss_scale(vsa, lp_train_everywhere, case):
    """``set_train_dtype("fp16")`` - the reference's own autocast type (train.py:120) - used the way the reference uses it,
    under a loss scale (GradScaler, train.py:60,126-128): loss and every unscaled gradient against the float64 goldens of the
    IMPORTED reference at tests/tolerances.py's TRAIN_FP16_*: 11 significant bits instead of bf16's 8, and the bounds are
    that much tighter than TRAIN_LP_*."""
    c = case
    z = np.load(os.path.join(GOLDEN, c["name"] + ".npz"))
    sd = vsa.synth.make_state_dict(c["d"], c["L"], c["wseed"])
    x, mask, target, R = _inputs(vsa.synth, c)
    m = vsa.SimNet(num_heads=c["H"], d_model=c["d"], num_layers=c["L"], sparsity=0.0, dropout=0.0)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).train().set_train_dtype("fp16")
    xd = x.to(_dev()).requires_grad_(True)
    md = None if mask is None else mask.to(_dev())
    pred, hidden = m(xd, md)
    assert m.last_train_dtype == "fp16"
    mk = md if md is not None else torch.zeros(x.shape[:2], dtype=torch.bool, device=_dev())
    loss = vsa.mse_with_mask_loss(pred, target.to(_dev()), mk)
    if c["hidden_w"]:
        loss = loss + c["hidden_w"] * (hidden * R.to(_dev())).sum()
    S = tol.TRAIN_FP16_LOSS_SCALE
    (loss * S).backward()
    torch.cuda.synchronize()
    want = float(z["loss"])
    assert abs(loss.item() - want) <= tol.TRAIN_FP16_LOSS_RTOL * max(1.0, abs(want)), (loss.item(), want)
    grads = {"x": xd.grad}
    grads.update({k: p.grad for k, p in m.named_parameters()})
    worst, worst_k, worst_l2, worst_l2k = 0.0, None, 0.0, None
    for k in json.loads(str(z["keys"])):
        g = grads[k]
        assert g is not None and torch.isfinite(g).all(), k
        g = g / S
        g2 = g.reshape(-1, g.shape[-1]) if g.dim() > 1 else g.reshape(1, -1)
        rows = torch.from_numpy(z["r:" + k])
        want_g = torch.from_numpy(z["g:" + k]).double()
        tot, nrm, gmax, ref32 = z["s:" + k]
        diff = g2[rows.to(g2.device)].double().cpu() - want_g
        err = diff.abs().max().item()
        l2 = diff.norm().item() / (want_g.norm().item() + 1e-30)
        if gmax < 1e-6:       # analytically zero (k.bias)
            assert g.double().norm().item() <= tol.TRAIN_FP16_ZERO_ATOL, "%s: |g| %.3e" % (k, g.double().norm().item())
            continue
        assert err <= tol.TRAIN_FP16_GRAD_RTOL * gmax + 1e-7, "%s: err %.3e, max|g| %.3e" % (k, err, gmax)
        fc1 = k.endswith("mlp.fc1.weight") or k.endswith("mlp.fc1.bias")
        assert l2 <= (tol.TRAIN_FP16_FC1_L2 if fc1 else tol.TRAIN_FP16_GRAD_L2), "%s: relative L2 error %.3e" % (k, l2)
        if err / gmax > worst:
            worst, worst_k = err / gmax, k
        if l2 > worst_l2:
            worst_l2, worst_l2k = l2, k
    assert worst > 1e-5, "the low-precision path did not run (gradients at exact-fp32 accuracy)"
    print("%s: fp16 training (loss scale %g), worst gradient error relative to the tensor's max: %.2e (%s); worst relative L2 error %.2e (%s)"
          % (c["name"], S, worst, worst_k, worst_l2, worst_l2k))


def test_fp16_training_overflow_reaches_gradscaler(vsa, lp_train_everywhere):
    """What the fp16 range costs, and that it fails the way the reference's loop expects (train.py:126-128:
    ``scaler.scale(loss).backward(); scaler.step(optim); scaler.update()``): with a loss scale that pushes the score
    gradient past 65 504 the operands round to inf, the gradients come out non-finite, GradScaler skips the step (the
    parameters do not move) and backs the scale off; at a sane scale the same loop steps."""
    torch.manual_seed(3)
    sd = vsa.synth.make_state_dict(256, 2, 31)
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=2, sparsity=0.0, dropout=0.1)
    m.load_state_dict(sd)
    m = m.to(_dev()).train().set_train_dtype("fp16")
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    x = vsa.synth.make_features(4, 200, 9, "pool5", [200, 150, 180, 120]).to(_dev())
    mask = vsa.synth.padding_mask(x)
    tgt = torch.rand(4, 200, generator=torch.Generator().manual_seed(3)).to(_dev())

    def step(scaler):
        before = [p.detach().clone() for p in m.parameters()]
        pred, _h = m(x, mask)
        loss = vsa.mse_with_mask_loss(pred, tgt, mask)
        opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        finite = all(bool(torch.isfinite(p.grad).all()) for p in m.parameters())
        scaler.step(opt)
        scaler.update()
        moved = any(not torch.equal(a, p.detach()) for a, p in zip(before, m.parameters()))
        return finite, moved

    big = torch.amp.GradScaler("cuda", init_scale=2.0 ** 40)
    finite, moved = step(big)
    assert not finite and not moved and big.get_scale() == 2.0 ** 39, (finite, moved, big.get_scale())
    sane = torch.amp.GradScaler("cuda", init_scale=2.0 ** 10)
    finite, moved = step(sane)
    assert finite and moved and sane.get_scale() == 2.0 ** 10, (finite, moved, sane.get_scale())


def test_fp16_training_loss_curve_tracks_the_exact_path(vsa, lp_train_everywhere):
    """The reference's loop shape (autocast + GradScaler, train.py:118-128) for 40 Adam steps, exact fp32 against the fp16 mode
    from one initialisation and one dropout seed stream: every step's loss within 1 % of the exact run's."""
    def run(dtype):
        torch.manual_seed(7)
        sd = vsa.synth.make_state_dict(256, 2, 31)
        m = vsa.SimNet(num_heads=4, d_model=256, num_layers=2, sparsity=0.0, dropout=0.1)
        m.load_state_dict(sd)
        m = m.to(_dev()).train().set_train_dtype(dtype)
        opt = torch.optim.Adam(m.parameters(), lr=2e-4)
        scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 12, enabled=dtype == "fp16")
        x = vsa.synth.make_features(4, 200, 9, "pool5", [200, 150, 180, 120]).to(_dev())
        mask = vsa.synth.padding_mask(x)
        tgt = torch.rand(4, 200, generator=torch.Generator().manual_seed(3)).to(_dev())
        out = []
        for _ in range(40):
            pred, _h = m(x, mask)
            loss = vsa.mse_with_mask_loss(pred, tgt, mask)
            opt.zero_grad(set_to_none=True)
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            out.append(loss.item())
        return out
    a, b = run("fp32"), run("fp16")
    assert a[-1] < 0.7 * a[0] and b[-1] < 0.7 * b[0], (a[0], a[-1], b[0], b[-1])
    rel = max(abs(u - v) / u for u, v in zip(a, b))
    print("loss curves: exact %.4f -> %.4f, fp16 %.4f -> %.4f, worst step-wise relative difference %.2e" % (a[0], a[-1], b[0], b[-1], rel))
    assert rel < 1e-2 and a != b


def test_bf16_training_loss_curve_tracks_the_exact_path(vsa, lp_train_everywhere):
    """40 Adam steps from one initialisation and one dropout seed stream, exact fp32 vs bf16 GEMMs: the losses fall
    together (every step within 3 % of the exact run's loss) and both end below 0.7 x their start."""
    def run(dtype):
        torch.manual_seed(7)
        sd = vsa.synth.make_state_dict(256, 2, 31)
        m = vsa.SimNet(num_heads=4, d_model=256, num_layers=2, sparsity=0.0, dropout=0.1)
        m.load_state_dict(sd)
        m = m.to(_dev()).train().set_train_dtype(dtype)
        opt = torch.optim.Adam(m.parameters(), lr=2e-4)
        x = vsa.synth.make_features(4, 200, 9, "pool5", [200, 150, 180, 120]).to(_dev())
        mask = vsa.synth.padding_mask(x)
        tgt = torch.rand(4, 200, generator=torch.Generator().manual_seed(3)).to(_dev())
        out = []
        for _ in range(40):
            pred, _h = m(x, mask)
            loss = vsa.mse_with_mask_loss(pred, tgt, mask)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            out.append(loss.item())
        return out
    a, b = run("fp32"), run("bf16")
    assert a[-1] < 0.7 * a[0] and b[-1] < 0.7 * b[0], (a[0], a[-1], b[0], b[-1])
    rel = max(abs(u - v) / u for u, v in zip(a, b))
    print("loss curves: exact %.4f -> %.4f, bf16 %.4f -> %.4f, worst step-wise relative difference %.2e" % (a[0], a[-1], b[0], b[-1], rel))
    assert rel < 3e-2 and a != b


def test_dropout_hash_statistics(vsa):
    """Keep rate, independence across modules (sites) and seeds, and determinism of the counter hash."""
    lib = vsa._lib.load()
    M, cols, p = 512, 1024, 0.3

    def draw(seed, site, pp=p):
        k = torch.empty(M, cols, dtype=torch.uint8, device=_dev())
        vsa._lib.check(lib.vs_train_dropout_mask_rows(k.data_ptr(), M, cols, seed, site, pp, _stream()))
        return k.cpu().double()

    a, a2, b, c = draw(11, 3), draw(11, 3), draw(11, 4), draw(12, 3)
    assert torch.equal(a, a2)
    n = a.numel()
    sigma = math.sqrt(p * (1 - p) / n)
    for k in (a, b, c):
        assert abs(k.mean().item() - (1 - p)) < 5 * sigma
    # rows, columns and the three draws are uncorrelated (|corr| ~ 1/sqrt(n))
    for u, v in ((a, b), (a, c), (a[:, :-1], a[:, 1:]), (a[:-1], a[1:])):
        cu, cv = u - u.mean(), v - v.mean()
        corr = (cu * cv).mean().item() / (cu.std().item() * cv.std().item())
        assert abs(corr) < 6 / math.sqrt(n), corr
    assert abs(a.mean(0).std().item() - math.sqrt(p * (1 - p) / M)) < 0.2 * math.sqrt(p * (1 - p) / M)     # per-column rates spread as binomial
    assert draw(5, 1, 0.0).min().item() == 1.0
    assert abs(draw(5, 1, 0.9).mean().item() - 0.1) < 5 * math.sqrt(0.09 / n)


# ---------------------------------------------------------------------------------------------
# whole model WITH dropout: explicit-mask float64 model fed with the library's masks
# ---------------------------------------------------------------------------------------------
def _hip_gates(vsa, module, out_tensor, B, T, d, L, bf16=False):
    """The ReLU-and-dropout gate the HIP backward applies, per layer: sign pattern of the MLP activation in the
    activation record the forward left for its backward (vs_train_saved_field).  bf16: the record was written by a
    forward whose GEMMs ran in the bf16 training mode - the activation is stored as bf16 (first half of its field)."""
    lib = vsa._lib.load()
    saved = out_tensor.grad_fn.saved_tensors[2]
    handle = module._packed.handle
    dl = getattr(module, "_lib_d", d)        # an embedded model's record has the library's (padded) width; the MLP axis keeps its index
    gates = {}
    for l in range(L):
        off, cnt = C.c_size_t(), C.c_size_t()
        vsa._lib.check(lib.vs_train_saved_field(handle, B, T, l, 0, C.byref(off), C.byref(cnt)))
        if bf16:        # True / "bf16": bf16 planes; "fp16": the fp16 training mode's
            act = saved[off.value: off.value + 2 * cnt.value].view(torch.float16 if bf16 == "fp16" else torch.bfloat16).view(B, T, 4 * dl).float()
        else:
            act = saved[off.value: off.value + 4 * cnt.value].view(torch.float32).view(B, T, 4 * dl)
        gates["gate%d" % l] = (act[..., :4 * d] > 0).cpu()
    return gates


def _library_masks(vsa, B, T, d, H, L, seed, p, p_embed):
    lib = vsa._lib.load()
    M = B * T

    def rows(site, cols, pp):
        k = torch.empty(M, cols, dtype=torch.uint8, device=_dev())
        vsa._lib.check(lib.vs_train_dropout_mask_rows(k.data_ptr(), M, cols, seed, site, pp, _stream()))
        return k.cpu().view(B, T, cols)

    masks = {}
    if p_embed > 0:
        masks["embed"] = rows(lib.vs_train_dropout_site(-1, 0), d, p_embed)
    for l in range(L):
        k = torch.empty(B, H, T, T, dtype=torch.uint8, device=_dev())
        vsa._lib.check(lib.vs_train_dropout_mask_attention(k.data_ptr(), B, H, T, seed, lib.vs_train_dropout_site(l, 0), p, _stream()))
        masks["attn%d" % l] = k.cpu()
        masks["drop1_%d" % l] = rows(lib.vs_train_dropout_site(l, 1), d, p)
        masks["mlp%d" % l] = rows(lib.vs_train_dropout_site(l, 2), 4 * d, p)
        masks["drop2_%d" % l] = rows(lib.vs_train_dropout_site(l, 3), d, p)
    return masks


@pytest.mark.parametrize("H,d,L,B,T,p,p_embed,masked", [(4, 256, 2, 2, 90, 0.3, 0.0, True), (8, 256, 1, 1, 130, 0.2, 0.5, False),
                                                        (4, 512, 1, 2, 70, 0.3, 0.0, True)])
def test_training_step_with_dropout_matches_explicit_mask_model(vsa, H, d, L, B, T, p, p_embed, masked):
    synth = vsa.synth
    sd = synth.make_state_dict(d, L, 21)
    m = vsa.SimNet(num_heads=H, d_model=d, num_layers=L, sparsity=p_embed, dropout=p)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).train()
    x = synth.make_features(B, T, 22, "randn", [T, T - 23][:B] if masked else None)
    mask = synth.padding_mask(x) if masked else None
    target = torch.rand(B, T, generator=torch.Generator().manual_seed(1))
    torch.manual_seed(77)
    seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())        # what _forward_train will draw
    torch.manual_seed(77)
    xd = x.to(_dev()).requires_grad_(True)
    md = None if mask is None else mask.to(_dev())
    pred, hidden = m(xd, md)
    mk = md if md is not None else torch.zeros(B, T, dtype=torch.bool, device=_dev())
    loss = vsa.mse_with_mask_loss(pred, target.to(_dev()), mk) + 1e-3 * hidden.sum()
    # the checker shares the implementation's ReLU-and-dropout gate (a ReLU input within fp32 rounding of zero may fall
    # on the other side in float64): read it from the forward's activation record BEFORE backward frees it
    gates = _hip_gates(vsa, m, pred, B, T, d, L)
    loss.backward()
    torch.cuda.synchronize()
    # float64 model with the library's masks
    masks = _library_masks(vsa, B, T, d, H, L, seed, p, p_embed)
    params = {k: v.double().clone().requires_grad_(k != "embedding_layer.positional_encoding.pos_embedding") for k, v in sd.items()}
    x64 = x.double().clone().requires_grad_(True)
    rl, rh = torch_ref.forward_with_masks(params, x64, mask, H, p, p_embed, masks, None, gates)
    scale = torch.ones(B, T, dtype=torch.float64) if mask is None else (~mask).double()
    rloss = (((rl.squeeze(2) - target.double()) * scale) ** 2).mean() + 1e-3 * rh.sum()
    rloss.backward()
    valid = torch.ones(B, T, dtype=torch.bool) if mask is None else ~mask
    assert (pred.detach().cpu().double() - rl.detach())[valid].abs().max().item() < 1e-4
    assert abs(loss.item() - rloss.item()) < 1e-5 * max(1.0, abs(rloss.item()))
    _close(xd.grad, x64.grad, "dx")
    for k, prm in m.named_parameters():
        _close(prm.grad, params[k].grad, k)
    # and dropout really happened: the same call in eval mode differs
    with torch.no_grad():
        e, _ = m.eval()(x.to(_dev()), md)
    assert (e - pred.detach())[valid.to(_dev())].abs().max().item() > 1e-3


def test_train_loop_like_the_reference(vsa):
    """train.py:111-131 verbatim shape: autocast, GradScaler, Adam, masked MSE; the loss must fall, and two runs from
    the same torch seed are bit-identical (dropout seeds come from torch's generator; every reduction is ordered)."""
    synth = vsa.synth

    def run():
        torch.manual_seed(1234)                                                   # utils.set_seed
        m = vsa.SimNet(num_heads=4, d_model=256, num_layers=2, sparsity=0.0, use_cls=False, dropout=0.3, num_classes=1,
                       use_pos=True).to(_dev())
        m.load_state_dict(synth.make_state_dict(256, 2, 3))
        opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=0.01)
        scaler = torch.amp.GradScaler("cuda")
        feature = synth.make_features(4, 120, 8, "pool5", [120, 100, 77, 51]).to(_dev())
        target = torch.rand(4, 120, generator=torch.Generator().manual_seed(2)).to(_dev())
        m.train()
        losses = []
        for _ in range(12):
            mask = (feature[:, :, 0] == 1000)                                     # train.py:118
            with torch.amp.autocast("cuda"):
                pred, _h = m(feature, mask)
                loss = vsa.mse_with_mask_loss(pred, target, mask)
            opt.zero_grad()
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            losses.append(loss.item())
        return losses, [p.detach().clone() for p in m.parameters()]

    l1, p1 = run()
    l2, p2 = run()
    assert all(math.isfinite(v) for v in l1) and min(l1[-3:]) < l1[0]
    assert l1 == l2 and all(torch.equal(a, b) for a, b in zip(p1, p2))


def test_mse_with_mask_loss_matches_reference_formula(vsa):
    g = torch.Generator().manual_seed(4)
    out = torch.randn(3, 50, 1, generator=g)
    tgt = torch.rand(3, 50, generator=g)
    mask = torch.zeros(3, 50, dtype=torch.bool)
    mask[1, 30:] = True
    mask[2, 10:] = True
    for reduction in ("avg", "sum"):
        o64 = out.double().clone().requires_grad_(True)
        sc = torch.ones(3, 50, dtype=torch.float64)
        sc[mask] = 0.0                                                             # utils.py:47-48
        want = ((o64.squeeze(2) * sc - tgt.double() * sc) ** 2)
        want = want.mean() if reduction == "avg" else want.sum()
        want.backward()
        od = out.to(_dev()).requires_grad_(True)
        got = vsa.mse_with_mask_loss(od, tgt.to(_dev()), mask.to(_dev()), reduction)
        (got * 3.0).backward()
        assert abs(got.item() - want.item()) < 1e-6 * max(1.0, want.item())
        assert (od.grad.cpu().double() - 3.0 * o64.grad).abs().max().item() < 1e-6


def test_embedded_shape_trains_like_a_native_one(vsa):
    """A model outside the kernels' envelope (8 heads of 16, d_model 128: run embedded in 8 x 32 = 256) through the reference's
    loop shape: dropout on, Adam steps (every step re-packs the zero-padded parameters), loss falls; per seed the step is
    bitwise reproducible; gradients have the parameters' TRUE shapes; with dropout off the autograd forward equals the
    scoring forward bit for bit where both run the same kernels' arithmetic (1e-5)."""
    def run():
        torch.manual_seed(11)
        sd = vsa.synth.make_state_dict(128, 2, 41)
        m = vsa.SimNet(num_heads=8, d_model=128, num_layers=2, sparsity=0.0, dropout=0.2)
        m.load_state_dict(sd)
        m = m.to(_dev()).train()
        opt = torch.optim.Adam(m.parameters(), lr=3e-4)
        x = vsa.synth.make_features(3, 120, 9, "pool5", [120, 80, 100]).to(_dev())
        mask = vsa.synth.padding_mask(x)
        tgt = torch.rand(3, 120, generator=torch.Generator().manual_seed(3)).to(_dev())
        losses = []
        for _ in range(30):
            pred, hidden = m(x, mask)
            assert hidden.shape == (3, 120, 128)
            loss = vsa.mse_with_mask_loss(pred, tgt, mask) + 1e-4 * hidden.square().mean()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            for p_ in m.parameters():
                assert p_.grad.shape == p_.shape and torch.isfinite(p_.grad).all()
            opt.step()
            losses.append(loss.item())
        return m, x, mask, losses
    m, x, mask, a = run()
    _m2, _x, _mk, b = run()
    assert a == b, "same seed, different losses"
    assert a[-1] < 0.7 * a[0], (a[0], a[-1])
    m.eval()
    with torch.no_grad():
        sl, sh = m(x, mask)
    xg = x.clone().requires_grad_(True)
    tl, th = m(xg, mask)                      # eval mode under autograd: the training kernels with dropout off
    valid = ~mask
    assert (tl - sl)[valid].abs().max().item() < 1e-5 and (th - sh)[valid].abs().max().item() < 1e-5
    th.sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_low_precision_qk_gradients_at_peaked_attention(vsa, lp_train_everywhere, dtype):
    """the soak normalises the q / k projection gradients by the v projection's (at DIFFUSE attention dS = P (dP -
    delta) is a near-cancelling difference and the operand rounding enters at the scale of dP), so a sign or scale bug confined
    to dS / dq / dk could hide there.  Here the attention is PEAKED (q / k weights scaled up 8x: the softmax puts most of a row
    on a few keys, dS is no longer a small difference) and the q / k gradients are held UN-normalised: cosine similarity with
    the float64 truth >= 0.995 and relative L2 <= 6e-2 (bf16; measured 3.0e-2) / 1e-2 (fp16; measured 4.5e-3), for the weights
    and the biases of q and k."""
    H, d, L = 4, 256, 2
    sd = vsa.synth.make_state_dict(d, L, 51)
    for l in range(L):
        for n in ("q", "k"):
            sd["encoder.module_list.%d.sa.%s.weight" % (l, n)] = sd["encoder.module_list.%d.sa.%s.weight" % (l, n)] * 8.0
    x = vsa.synth.make_features(2, 160, 17, "randn")
    target = torch.rand(2, 160, generator=torch.Generator().manual_seed(4))
    # float64 truth (dropout off: train-mode forward of the restatement, masks=None)
    params = {k: v.double().clone().requires_grad_(v.dtype.is_floating_point and "pos_embedding" not in k) for k, v in sd.items()}
    pl, _ph = torch_ref.forward_with_masks(params, x.double(), None, H)
    ((pl.squeeze(2) - target.double()) ** 2).mean().backward()
    # how peaked: the mean largest attention probability of layer 0 (for the record)
    m = vsa.SimNet(num_heads=H, d_model=d, num_layers=L, sparsity=0.0, dropout=0.0)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).train().set_train_dtype(dtype)
    pred, _hid = m(x.to(_dev()), None)
    S = tol.TRAIN_FP16_LOSS_SCALE if dtype == "fp16" else 1.0
    (vsa.mse_with_mask_loss(pred, target.to(_dev()), torch.zeros(2, 160, dtype=torch.bool, device=_dev())) * S).backward()
    assert m.last_train_dtype == dtype
    worst_cos, worst_l2 = 1.0, 0.0
    for l in range(L):
        for n in ("q", "k"):
            for part in ("weight", "bias"):
                key = "encoder.module_list.%d.sa.%s.%s" % (l, n, part)
                want = params[key].grad.flatten()
                got = (dict(m.named_parameters())[key].grad / S).double().cpu().flatten()
                if want.norm().item() < 1e-9:          # k.bias: analytically zero (softmax shift invariance)
                    continue
                cos = torch.dot(got, want).item() / (got.norm().item() * want.norm().item() + 1e-300)
                l2 = (got - want).norm().item() / want.norm().item()
                worst_cos, worst_l2 = min(worst_cos, cos), max(worst_l2, l2)
                assert cos >= 0.995, "%s: cosine %.4f" % (key, cos)
                assert l2 <= (6e-2 if dtype == "bf16" else 1e-2), "%s: relative L2 %.3e" % (key, l2)
    print("%s, peaked attention: q / k gradients worst cosine %.5f, worst relative L2 %.2e" % (dtype, worst_cos, worst_l2))
