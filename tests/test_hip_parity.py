"""GPU parity tests: the HIP path (through the C ABI of libvsscore.so) against the committed
reference-generated golden vectors and against the CPU oracle on the same seeded inputs.

Tolerance: 1e-4 absolute on fp32 logits / hidden state (BASELINE.json north_star)."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import build_case, golden_cases, load_golden
import tolerances as tol
from oracle.simnet_oracle import oracle_forward

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _dev():
    assert torch.cuda.is_available(), "these tests need a HIP device"
    return torch.device("cuda:0")


def _stream():
    return torch.cuda.current_stream().cuda_stream


_F64 = {}


def _fp64_goldens():
    if not _F64:
        import json
        import os
        import numpy as np
        from conftest import GOLDEN
        _F64["npz"] = np.load(os.path.join(GOLDEN, "forward_fp64.npz"))
        with open(os.path.join(GOLDEN, "forward_fp64.json")) as f:
            _F64["meta"] = json.load(f)
    return _F64


def _model(vsa, c, sd):
    m = vsa.SimNet(num_heads=c["H"], d_model=c["d"], num_layers=c["L"], sparsity=0.0, dropout=0.3,
                   num_classes=c.get("num_classes", 1), use_pos=c.get("use_pos", True))
    m.load_state_dict(sd, strict=True)
    return m.to(_dev()).eval()


@pytest.fixture(params=["auto", "tiled"])
def kernel_path(request, vsa):
    """Small inputs take the skinny latency kernels by default; VS_SKINNY_ROWS=0 (vs_set_option) pins the
    LDS-tiled throughput kernels, so both families are held to the same vectors."""
    if request.param != "auto":
        vsa._lib.set_option("VS_SKINNY_ROWS", 0)
    yield request.param
    vsa._lib.set_option("VS_SKINNY_ROWS", -1)


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_forward_matches_reference_golden(vsa, case, kernel_path):
    """Full scorer forward vs vectors produced by the imported reference (tests/golden/make_golden.py)."""
    g = load_golden(case["name"])
    sd, x, mask = build_case(vsa.synth, case)
    model = _model(vsa, case, sd)
    with torch.no_grad():
        logits, hidden = model(x.to(_dev()), None if mask is None else mask.to(_dev()))
        logits2, inter = model(x.to(_dev()), None if mask is None else mask.to(_dev()), model_score=True)
    torch.cuda.synchronize()
    assert logits.shape == g["logits"].shape and hidden.shape[:2] == x.shape[:2]
    valid = torch.ones(x.shape[:2], dtype=torch.bool) if mask is None else ~mask
    # padded QUERY rows are computed but never read by the callers (utils.py:47-51); compare valid frames
    dl = (logits.cpu() - g["logits"])[valid].abs().max().item()
    dh = (hidden.cpu()[:, g["rows"]] - g["hidden"])[valid[:, g["rows"]]].abs().max().item()
    assert dl < TOL and dh < TOL, (dl, dh)
    assert torch.equal(logits, logits2) and torch.equal(hidden, inter)
    # float64 truth (tests/golden/make_golden_fp64.py: the reference run in double): the HIP path's distance to it,
    # beside the distance of the reference's own fp32 run (the primary golden above) to it
    f64 = _fp64_goldens()
    l64, h64 = torch.from_numpy(f64["npz"][case["name"] + ":logits"]), torch.from_numpy(f64["npz"][case["name"] + ":hidden"])
    dl64 = (logits.cpu().double() - l64)[valid].abs().max().item()
    dh64 = (hidden.cpu()[:, g["rows"]].double() - h64)[valid[:, g["rows"]]].abs().max().item()
    ref = f64["meta"]["cases"][case["name"]]
    print("%s [%s]: HIP vs fp64 logits %.2e hidden %.2e | reference fp32 vs fp64 logits %.2e hidden %.2e | HIP vs fp32 golden %.2e %.2e"
          % (case["name"], kernel_path, dl64, dh64, ref["ref32_vs_ref64_logits"], ref["ref32_vs_ref64_hidden"], dl, dh))
    assert dl64 < TOL and dh64 < TOL, (dl64, dh64)


def test_padded_query_rows_match_oracle_too(vsa):
    """Even the padded query rows (finite garbage in the reference) agree with the oracle."""
    synth = vsa.synth
    sd = synth.make_state_dict(256, 4, 21)
    x = synth.make_features(2, 140, 22, "randn", [140, 77])
    mask = synth.padding_mask(x)
    c = dict(H=4, d=256, L=4)
    with torch.no_grad():
        logits, hidden = _model(vsa, c, sd)(x.to(_dev()), mask.to(_dev()))
        rl, rh = oracle_forward(sd, x, mask, 4)
    # pad rows carry |x| = 1000 features, so their activations are large: relative bound
    assert torch.allclose(logits.cpu(), rl, atol=TOL, rtol=1e-4)
    assert torch.allclose(hidden.cpu(), rh, atol=TOL, rtol=1e-4)


def test_non_tensor_mask_is_ignored(vsa):
    """train.py:162 passes mask=True; the reference ignores non-Tensor masks (simnet.py:38)."""
    synth = vsa.synth
    sd = synth.make_state_dict(256, 2, 5)
    x = synth.make_features(1, 70, 6).to(_dev())
    m = _model(vsa, dict(H=4, d=256, L=2), sd)
    with torch.no_grad():
        a, _ = m(x, True)
        b, _ = m(x)
    assert torch.equal(a, b)


def test_fused_sigmoid_and_score(vsa):
    synth = vsa.synth
    sd = synth.make_state_dict(256, 2, 8)
    x = synth.make_features(2, 100, 9)
    m = _model(vsa, dict(H=4, d=256, L=2), sd)
    with torch.no_grad():
        logits, _ = m(x.to(_dev()))
        s = m.score(x.to(_dev()))
        rl, _ = oracle_forward(sd, x, None, 4)
    assert (s.cpu() - torch.sigmoid(rl.squeeze(-1))).abs().max().item() < TOL
    assert (torch.sigmoid(logits.squeeze(-1)) - s).abs().max().item() < 1e-6


def test_repack_after_parameter_update(vsa):
    """Packed device weights must follow optimizer steps / load_state_dict (SURVEY §8(b) ownership)."""
    synth = vsa.synth
    sd1, sd2 = synth.make_state_dict(256, 2, 31), synth.make_state_dict(256, 2, 32)
    x = synth.make_features(1, 64, 33)
    m = _model(vsa, dict(H=4, d=256, L=2), sd1)
    with torch.no_grad():
        a, _ = m(x.to(_dev()))
        m.load_state_dict(sd2)
        b, _ = m(x.to(_dev()))
        m.final_layer.bias.add_(1.0)
        c, _ = m(x.to(_dev()))
        rb, _ = oracle_forward(sd2, x, None, 4)
    assert (b.cpu() - rb).abs().max().item() < TOL
    assert (a - b).abs().max().item() > 1e-3
    assert (c - b - 1.0).abs().max().item() < 1e-5


def test_errors_mirror_reference(vsa):
    synth = vsa.synth
    m = _model(vsa, dict(H=4, d=256, L=1), synth.make_state_dict(256, 1, 1))
    with torch.no_grad():
        with pytest.raises(RuntimeError):          # T > 2000: reference raises on the PE add
            m(torch.zeros(1, 2001, 1024, device=_dev()))
        with pytest.raises(RuntimeError):          # D != 1024
            m(torch.zeros(1, 10, 2048, device=_dev()))
        with pytest.raises(RuntimeError):          # CPU tensors: no fallback
            m.cpu()(torch.zeros(1, 10, 1024))


# ---------------------------------------------------------------------------------------------
# per-kernel parity through the exported C entry points
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("M,N,K,relu,T", [(300, 256, 1024, 0, 0), (129, 1024, 256, 1, 0), (64, 768, 256, 0, 0),
                                          (1000, 256, 1024, 0, 250), (37, 512, 2048, 0, 37), (2048, 2048, 512, 1, 0)])
def test_linear_kernel(vsa, M, N, K, relu, T, kernel_path):
    lib = vsa._lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    pe = torch.randn(T, N, generator=g) if T else None
    ref = F.linear(A.double(), W.double(), b.double())
    if relu:
        ref = F.relu(ref)
    if T:
        ref = ref + pe.double().repeat(M // T, 1)
    dA, dW, db = A.to(_dev()), W.to(_dev()), b.to(_dev())
    dpe = pe.to(_dev()) if T else None
    out = torch.full((M, N), float("nan"), device=_dev())
    vsa._lib.check(lib.vs_linear_f32(dA.data_ptr(), dW.data_ptr(), db.data_ptr(), out.data_ptr(), M, N, K, relu,
                                     dpe.data_ptr() if T else None, T, _stream()))
    torch.cuda.synchronize()
    assert (out.cpu().double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,T,d,H", [(2, 100, 256, 4), (1, 33, 512, 4), (3, 64, 256, 8)])
def test_qkv_kernel(vsa, B, T, d, H, kernel_path):
    lib = vsa._lib.load()
    g = torch.Generator().manual_seed(B * T + d)
    h = torch.randn(B, T, d, generator=g)
    W = torch.randn(3 * d, d, generator=g) / math.sqrt(d)
    b = torch.randn(3 * d, generator=g)
    ref = F.linear(h.double(), W.double(), b.double()).view(B, T, 3, H, d // H).permute(2, 0, 3, 1, 4)
    out = torch.full((3, B, H, T, d // H), float("nan"), device=_dev())
    dh_, dW, db = h.to(_dev()), W.to(_dev()), b.to(_dev())
    vsa._lib.check(lib.vs_qkv_proj_f32(dh_.data_ptr(), dW.data_ptr(), db.data_ptr(), out.data_ptr(), B, T, d, H, _stream()))
    torch.cuda.synchronize()
    assert (out.cpu().double() - ref).abs().max().item() < 2e-5


def _attn_ref(q, k, v, mask, scale):
    s = torch.matmul(q.double(), k.double().transpose(2, 3)) * scale
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    o = torch.matmul(torch.softmax(s, dim=3), v.double())
    B, H, T, dh = q.shape
    return o.permute(0, 2, 1, 3).reshape(B, T, H * dh)


@pytest.mark.parametrize("B,H,T,dh,masked", [(2, 4, 320, 64, False), (1, 4, 1024, 64, False), (2, 4, 200, 64, True),
                                             (2, 4, 150, 128, True), (1, 8, 65, 32, True), (1, 4, 31, 64, False),
                                             (1, 2, 129, 128, False)])
def test_attention_kernel(vsa, B, H, T, dh, masked):
    lib = vsa._lib.load()
    g = torch.Generator().manual_seed(T + dh)
    q, k, v = (torch.randn(B, H, T, dh, generator=g) * 2.0 for _ in range(3))
    mask = vsa.synth.random_mask(B, T, 3) if masked else None
    scale = (H * dh) ** -0.5
    ref = _attn_ref(q, k, v, mask, scale)
    dq, dk, dv = q.to(_dev()), k.to(_dev()), v.to(_dev())
    dm = mask.to(_dev()) if masked else None
    out = torch.full((B, T, H * dh), float("nan"), device=_dev())
    vsa._lib.check(lib.vs_attention_f32(dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dm.data_ptr() if masked else None,
                                        out.data_ptr(), B, H, T, dh, scale, _stream()))
    torch.cuda.synchronize()
    assert (out.cpu().double() - ref).abs().max().item() < 2e-5


def test_attention_online_softmax_rescale_branch(vsa):
    """Force the running-max jump: one key row spikes against every query in a LATE tile, so every
    block has to rescale its accumulated O and l (guide rule: a rare data-dependent branch needs its own test)."""
    lib = vsa._lib.load()
    B, H, T, dh = 1, 4, 512, 64
    g = torch.Generator().manual_seed(99)
    q, k, v = (torch.randn(B, H, T, dh, generator=g) for _ in range(3))
    k[:, :, 300] = q.mean(dim=2) * 50.0 + 20.0
    k[:, :, 77] = -k[:, :, 300]
    scale = 1.0
    ref = _attn_ref(q, k, v, None, scale)
    dq, dk, dv = q.to(_dev()), k.to(_dev()), v.to(_dev())
    out = torch.full((B, T, H * dh), float("nan"), device=_dev())
    vsa._lib.check(lib.vs_attention_f32(dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), None, out.data_ptr(), B, H, T, dh,
                                        scale, _stream()))
    torch.cuda.synchronize()
    assert (out.cpu().double() - ref).abs().max().item() < 5e-5


@pytest.mark.parametrize("M,N,K,nc,sig", [(300, 256, 256, 0, 0), (100, 256, 1024, 1, 0), (65, 512, 2048, 1, 1),
                                          (64, 128, 512, 3, 0), (1000, 320, 320, 2, 1), (33, 64, 256, 1, 0)])
def test_linear_residual_layernorm_kernel(vsa, M, N, K, nc, sig, kernel_path):
    lib = vsa._lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    gam, bet = 1 + 0.1 * torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g)
    sw, sb = torch.randn(max(nc, 1), N, generator=g) / math.sqrt(N), torch.randn(max(nc, 1), generator=g)
    y = F.layer_norm(F.linear(A.double(), W.double(), b.double()) + res.double(), (N,), gam.double(), bet.double(), 1e-5)
    sc = F.linear(y, sw.double(), sb.double())
    if sig:
        sc = torch.sigmoid(sc)
    d = [t.to(_dev()) for t in (A, W, b, res, gam, bet, sw, sb)]
    out = torch.full((M, N), float("nan"), device=_dev())
    scores = torch.full((M, max(nc, 1)), float("nan"), device=_dev())
    vsa._lib.check(lib.vs_linear_residual_layernorm_f32(
        d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), d[5].data_ptr(),
        out.data_ptr(), M, N, K, d[6].data_ptr() if nc else None, d[7].data_ptr() if nc else None, nc, sig,
        scores.data_ptr() if nc else None, _stream()))
    torch.cuda.synchronize()
    assert (out.cpu().double() - y).abs().max().item() < 2e-5
    if nc:
        assert (scores.cpu().double() - sc).abs().max().item() < 2e-5


def test_c_abi_rejects_bad_arguments(vsa):
    lib = vsa._lib.load()
    desc = vsa._lib.ModelDesc(100, 4, 1, 1024, 2000, 1)        # d_model not a multiple of 64
    out = C.c_void_p()
    P = vsa._lib.ModelParams()
    rc = lib.vs_weights_pack(C.byref(desc), C.byref(P), None, C.byref(out))
    assert rc == vsa._lib.VS_ERR_INVALID and b"d_model" in lib.vs_last_error()
    assert lib.vs_scorer_workspace_bytes(None, 1, 1) == 0


# ---------------------------------------------------------------------------------------------
# full-size properties (BASELINE.json configs[2]: B=64, T=1024) that need no CPU oracle run
# ---------------------------------------------------------------------------------------------

def test_full_size_batch_properties(vsa):
    """At B=64,T=1024 (bench size): (i) videos are independent — scoring a slice of the batch gives
    bit-identical rows; (ii) a padded+masked copy of a short video scores its valid frames like the
    unpadded video within fp32 noise; (iii) outputs are finite."""
    synth = vsa.synth
    sd = synth.make_state_dict(256, 4, 41)
    m = _model(vsa, dict(H=4, d=256, L=4), sd)
    g = torch.Generator(device="cpu").manual_seed(1234)
    x = torch.randn(64, 1024, 1024, generator=g).to(_dev())
    with torch.no_grad():
        full, hid = m(x)
        part, _ = m(x[5:9].contiguous())
        assert torch.isfinite(full).all() and torch.isfinite(hid).all()
        assert torch.equal(full[5:9], part)
        short = x[:2, :700].contiguous()
        padded = torch.full((2, 1024, 1024), 1000.0, device=_dev())
        padded[:, :700] = short
        a, _ = m(padded, padded[:, :, 0] == 1000.0)
        b, _ = m(short)
    assert (a[:, :700] - b).abs().max().item() < 2e-5
    # and the oracle on two of the 64 videos (CPU cost ~1 s)
    with torch.no_grad():
        rl, _ = oracle_forward(sd, x[:2].cpu(), None, 4)
    assert (full[:2].cpu() - rl).abs().max().item() < TOL


def test_corpus_scoring_is_sharding_invariant_bit_for_bit(vsa):
    """BASELINE configs[3] shape (ragged corpus, key masks on): scoring the corpus as 1, 2 or 8 shards
    gives bit-identical per-frame scores (videos are independent; the kernels are batch-invariant),
    and matches the oracle per video."""
    import importlib
    corpus = importlib.import_module("video-summarization_amd.corpus")
    synth = vsa.synth
    sd = synth.make_state_dict(256, 4, 51)
    m = _model(vsa, dict(H=4, d=256, L=4), sd)
    g = torch.Generator().manual_seed(7)
    lens = torch.randint(100, 650, (24,), generator=g).tolist()
    vids = [torch.randn(t, 1024, generator=g) for t in lens]
    fn = lambda x, mask: m.score(x, mask)
    base = corpus.score_corpus(fn, vids, device=_dev(), max_frames=4096)
    for world in (2, 8):
        for rank in range(world):
            mine = corpus.plan_shards(lens, world)[rank]
            part = corpus.score_corpus(fn, [vids[i] for i in mine], device=_dev(), max_frames=2048)
            for slot, i in enumerate(mine):
                assert torch.equal(part[slot], base[i])
    with torch.no_grad():
        for i in (0, 5, 23):
            rl, _ = oracle_forward(sd, vids[i].unsqueeze(0), None, 4)
            assert (base[i] - torch.sigmoid(rl[0, :, 0])).abs().max().item() < TOL


def test_long_video_shape_beyond_reference_envelope(vsa):
    """BASELINE configs[4] shape (T=8192, 2048-d features), fp32: needs in_features=2048 and an 8192-row
    positional table, both outside the reference's hard-coded envelope, so the checker is the
    oracle restatement re-parameterised the same way.  One video, M-A, 2 layers (CPU oracle
    materialises [1,4,8192,8192] per layer)."""
    synth = vsa.synth
    sd = synth.make_state_dict(256, 2, 61, in_features=2048, max_len=8192)
    x = synth.make_features(1, 8192, 62, "randn", in_features=2048)
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=2, sparsity=0.0, dropout=0.3, in_features=2048, pe_len=8192)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval()
    with torch.no_grad():
        logits, hidden = m(x.to(_dev()))
        torch.set_num_threads(16)
        rl, rh = oracle_forward(sd, x, None, 4)
    assert (logits.cpu() - rl).abs().max().item() < TOL
    assert (hidden.cpu() - rh).abs().max().item() < TOL


@pytest.mark.parametrize("T", [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 257])
def test_awkward_lengths_match_oracle(vsa, T, kernel_path):
    """Every tile/tail boundary of the kernels (32-row MFMA blocks, 64-key tiles, 128/256-row blocks), with
    a ragged padding mask and an odd batch size, through both kernel families."""
    synth = vsa.synth
    sd = synth.make_state_dict(256, 2, 70 + T)
    lens = [T, max(1, T // 2), max(1, T - 1)]
    x = synth.make_features(3, T, 71 + T, "randn", lens)
    mask = synth.padding_mask(x)
    m = _model(vsa, dict(H=4, d=256, L=2), sd)
    with torch.no_grad():
        logits, hidden = m(x.to(_dev()), mask.to(_dev()))
        nomask, _ = m(x[:1].to(_dev()))
        rl, rh = oracle_forward(sd, x, mask, 4)
        rn, _ = oracle_forward(sd, x[:1], None, 4)
    valid = ~mask
    assert (logits.cpu() - rl)[valid].abs().max().item() < TOL
    assert (hidden.cpu() - rh)[valid].abs().max().item() < TOL
    assert (nomask.cpu() - rn).abs().max().item() < TOL


@pytest.mark.parametrize("compute", ["fp32", "fp16x3", "bf16"])
def test_fully_masked_video_yields_nan_like_the_reference(vsa, compute):
    """softmax over an all-masked key row is NaN in the reference; same here (no crash, no hang),
    in every compute mode."""
    synth = vsa.synth
    sd = synth.make_state_dict(256, 1, 5)
    x = synth.make_features(2, 40, 6)
    mask = torch.zeros(2, 40, dtype=torch.bool)
    mask[1] = True
    m = _model(vsa, dict(H=4, d=256, L=1), sd).set_compute_dtype(compute)
    with torch.no_grad():
        logits, _ = m(x.to(_dev()), mask.to(_dev()))
        rl, _ = oracle_forward(sd, x, mask, 4)
    assert torch.isnan(rl[1]).all() and torch.isnan(logits[1].cpu()).all()
    assert (logits[0].cpu() - rl[0]).abs().max().item() < (TOL if compute != "bf16" else BF16_LOGIT_TOL)


def test_two_streams_and_two_models_do_not_interfere(vsa):
    """Weights handles and workspaces are per call/module: two models on two streams give the same scores as alone."""
    synth = vsa.synth
    sd1, sd2 = synth.make_state_dict(256, 2, 91), synth.make_state_dict(256, 2, 92)
    m1, m2 = _model(vsa, dict(H=4, d=256, L=2), sd1), _model(vsa, dict(H=4, d=256, L=2), sd2)
    x = synth.make_features(4, 300, 93).to(_dev())
    with torch.no_grad():
        a1, _ = m1(x)
        a2, _ = m2(x)
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        outs = []
        for _ in range(3):
            with torch.cuda.stream(s1):
                b1, _ = m1(x)
            with torch.cuda.stream(s2):
                b2, _ = m2(x)
            outs.append((b1, b2))
        torch.cuda.synchronize()
    for b1, b2 in outs:
        assert torch.equal(a1, b1) and torch.equal(a2, b2)


def test_val_step_end_to_end_matches_reference(vsa):
    """BASELINE configs[0]/SURVEY §8(c): the reference's val_step (train.py:134-152) on synthetic TVSum-shaped
    records of split 0 — reference model + reference evaluation produced the golden
    (tests/golden/make_golden_valstep.py).  Here: HIP scorer + C++ evaluation, per video and batched."""
    import importlib
    import os
    import sys
    import numpy as np
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    mk = importlib.import_module("make_golden_valstep")
    harness = importlib.import_module("video-summarization_amd.harness")
    g = np.load(os.path.join(GOLDEN, "valstep_golden.npz"))
    recs = mk.make_records()
    sd = vsa.synth.make_state_dict(256, 4, mk.WSEED)
    m = _model(vsa, dict(H=4, d=256, L=4), sd)
    loader = [(f.unsqueeze(0), t.unsqueeze(0), u) for f, t, u in recs]
    loss, f, k, s = harness.val_step(m, loader, _dev())
    with torch.no_grad():
        for feats, _, u in recs:
            sc = torch.sigmoid(m(feats.unsqueeze(0).to(_dev()))[0].view(-1)).cpu().numpy()
            assert np.abs(sc - g["scores_" + u.name]).max() < TOL
    assert abs(loss - float(g["loss"])) < 1e-5
    assert abs(f - g["metrics"][0]) < 1e-6 and abs(k - g["metrics"][1]) < 1e-4 and abs(s - g["metrics"][2]) < 1e-4
    lb, fb, kb, sb = harness.val_step_batched(m, [r[0] for r in recs], [r[1] for r in recs], [r[2] for r in recs], _dev())
    assert abs(lb - loss) < 1e-6 and abs(fb - f) < 1e-9 and abs(kb - k) < 1e-12 and abs(sb - s) < 1e-12


@pytest.fixture
def lp_linear_everywhere(vsa):
    """The forward keeps the exact fp32 latency kernels for batches of up to 8192 frames whatever the precision
    flags say; VS_LP_MIN_ROWS=0 pins the low-precision Linear kernels so small test batches exercise them."""
    vsa._lib.set_option("VS_LP_MIN_ROWS", 0)
    yield
    vsa._lib.set_option("VS_LP_MIN_ROWS", -1)


# ---- opt-in bf16 attention (VS_FLAG_BF16_ATTENTION; BASELINE configs[4] names bf16) ----------------------
# Tolerances, stated: the bf16 path rounds q*scale, k, v and the probabilities to 8-bit mantissas (relative
# 2^-9 each) before the two products and accumulates in fp32.  Per-kernel: |out - fp64 reference| <= 1.5e-2 of
# the largest |reference| entry (4e-3 against a checker that shares the rounded operands).  End to end
# (post-LN blocks renormalise every layer): logits within 2e-3 of the fp32 oracle and the sigmoid scores the
# summariser consumes within 5e-4 (measured on trained-like weights: 3.8e-4 and 9.5e-5).  The 1e-4 bar applies to the default
# fp32 path only.
BF16_ATTN_REL = tol.BF16_ATTN_KERNEL_REL
BF16_LOGIT_TOL = tol.BF16_ATTN_LOGIT_TOL        # attention alone on the bf16 pipe
BF16_SCORE_TOL = tol.BF16_ATTN_SCORE_TOL


def _attn_ref_bf16_operands(q, k, v, mask, scale):
    """fp64 attention over the operands as the kernel rounds them (q*scale*log2e, k, v to bf16): what is left
    between this and the kernel is the bf16 rounding of the probabilities and fp32 accumulation order."""
    rb = lambda t: t.to(torch.bfloat16).double()
    s2 = torch.matmul(rb(q * (scale * 1.4426950408889634)), rb(k).transpose(2, 3))
    if mask is not None:
        s2 = s2.masked_fill(mask[:, None, None, :], float("-inf"))
    p = torch.exp2(s2 - s2.max(dim=3, keepdim=True).values)
    o = torch.matmul(p, rb(v)) / p.sum(dim=3, keepdim=True)
    B, H, T, dh = q.shape
    return o.permute(0, 2, 1, 3).reshape(B, T, H * dh)


def _run_attn_bf16(vsa, q, k, v, mask, scale):
    lib = vsa._lib.load()
    B, H, T, dh = q.shape
    dq, dk, dv = q.to(_dev()), k.to(_dev()), v.to(_dev())
    dm = mask.to(_dev()) if mask is not None else None
    out = torch.full((B, T, H * dh), float("nan"), device=_dev())
    vsa._lib.check(lib.vs_attention_bf16(dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                         dm.data_ptr() if dm is not None else None, out.data_ptr(), B, H, T, dh,
                                         scale, _stream()))
    torch.cuda.synchronize()
    return out.cpu()


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,T,dh,masked", [(2, 4, 320, 64, False), (1, 4, 1024, 64, False), (2, 4, 200, 64, True),
                                             (1, 8, 65, 32, True), (1, 4, 31, 64, False), (3, 8, 257, 32, False),
                                             (1, 1, 1, 64, False),
                                             (2, 4, 320, 128, False), (1, 2, 700, 128, True), (1, 4, 1, 128, False)])
def test_attention_bf16_kernel(vsa, B, H, T, dh, masked):
    g = torch.Generator().manual_seed(T + dh)
    q, k, v = (torch.randn(B, H, T, dh, generator=g) * 2.0 for _ in range(3))
    mask = vsa.synth.random_mask(B, T, 3) if masked else None
    scale = (H * dh) ** -0.5
    ref = _attn_ref(q, k, v, mask, scale)
    out = _run_attn_bf16(vsa, q, k, v, mask, scale)
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() < BF16_ATTN_REL * ref.abs().max().item()
    ref_b = _attn_ref_bf16_operands(q, k, v, mask, scale)
    assert (out.double() - ref_b).abs().max().item() < 4e-3 * ref_b.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("T,dh", [(64, 64), (200, 64), (513, 64), (300, 32), (257, 128), (640, 128)])
def test_attention_bf16_operand_layout_is_exact_on_a_permutation(vsa, T, dh):
    """Every index map of both bf16 products, checked exactly: keys are +-1 vectors (exact in bf16) and query i is
    a scaled copy of key pi(i), so softmax row i is one-hot at pi(i) to ~e^-30 and the output must be
    V[pi(i)] (small integers, exact in bf16) — any wrong k-index, lane or key mapping shows up as a wrong row."""
    B, H = 2, 2
    g = torch.Generator().manual_seed(T)
    k = (torch.randint(0, 2, (B, H, T, dh), generator=g) * 2 - 1).float()
    # make keys pairwise distinguishable: stamp the key index in binary (as +-1) into the first 12 dims
    idx = torch.arange(T)
    bits = ((idx[:, None] >> torch.arange(12)[None, :]) & 1).float() * 2 - 1
    k[..., :12] = bits
    perm = torch.stack([torch.randperm(T, generator=g) for _ in range(B * H)]).view(B, H, T)
    q = torch.gather(k, 2, perm[..., None].expand(-1, -1, -1, dh))
    v = torch.randint(-8, 9, (B, H, T, dh), generator=g).float()
    scale = 1.0       # q.k = dh at the matching key, <= dh - 2 elsewhere... times 16 below
    out = _run_attn_bf16(vsa, q * 16.0, k, v, None, scale)
    want = torch.gather(v, 2, perm[..., None].expand(-1, -1, -1, dh)).permute(0, 2, 1, 3).reshape(B, T, H * dh)
    assert (out - want).abs().max().item() < 1e-5


@pytest.mark.gpu
def test_attention_bf16_rescale_branch(vsa):
    B, H, T, dh = 1, 4, 512, 64
    g = torch.Generator().manual_seed(99)
    q, k, v = (torch.randn(B, H, T, dh, generator=g) for _ in range(3))
    k[:, :, 300] = q.mean(dim=2) * 50.0 + 20.0
    k[:, :, 77] = -k[:, :, 300]
    # scores of several hundred: the input rounding alone moves the softmax, so the checker shares the rounded operands
    ref = _attn_ref_bf16_operands(q, k, v, None, 1.0)
    out = _run_attn_bf16(vsa, q, k, v, None, 1.0)
    assert (out.double() - ref).abs().max().item() < 4e-3 * ref.abs().max().item()


# ---- the bf16 mode's storage form (bf16 q * scale * log2 e / k / v planes in, bf16 out): head dim 64 runs on the
# one-wave-per-SIMD kernel (csrc/vs_attention_w64.hip), everything else on attn_fwd_lp_pipe<.., IO16> ----
def _stored_operands(q, k, v, scale):
    return ((q * (scale * 1.4426950408889634)).to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16))


def _attn_ref_stored(q16, k16, v16, mask):
    """fp64 attention over the stored bf16 operands (q16 already carries scale * log2 e)"""
    s2 = torch.matmul(q16.double(), k16.double().transpose(2, 3))
    if mask is not None:
        s2 = s2.masked_fill(mask[:, None, None, :], float("-inf"))
    p = torch.exp2(s2 - s2.max(dim=3, keepdim=True).values)
    o = torch.matmul(p, v16.double()) / p.sum(dim=3, keepdim=True)
    B, H, T, dh = q16.shape
    return o.permute(0, 2, 1, 3).reshape(B, T, H * dh)


def _run_attn_stored(vsa, q16, k16, v16, mask, w64=1, checked=0):
    lib = vsa._lib.load()
    B, H, T, dh = q16.shape
    dq, dk, dv = q16.to(_dev()), k16.to(_dev()), v16.to(_dev())
    dm = mask.to(_dev()) if mask is not None else None
    out = torch.full((B, T, H * dh), float("nan"), device=_dev(), dtype=torch.bfloat16)
    try:
        vsa._lib.set_option("VS_ATTN_W64", w64)
        vsa._lib.set_option("VS_ATTN_W64_CHECKED", checked)
        vsa._lib.check(lib.vs_attention_bf16_stored(dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                                    dm.data_ptr() if dm is not None else None, out.data_ptr(), B, H, T, dh,
                                                    _stream()))
        torch.cuda.synchronize()
    finally:
        vsa._lib.set_option("VS_ATTN_W64", -1)
        vsa._lib.set_option("VS_ATTN_W64_CHECKED", -1)
    return out.cpu()


BF16_STORED_REL = 6e-3      # against fp64 on the SAME stored operands: P and the output are rounded to bf16 (2^-9 each)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,T,masked", [(1, 1, 64, False), (2, 4, 320, False), (1, 4, 1024, False), (1, 2, 65, False),
                                          (1, 1, 1, False), (1, 4, 31, False), (2, 2, 513, False), (1, 2, 200, False),
                                          (1, 1, 2048, False), (3, 2, 1000, False), (1, 1, 257, False),
                                          (2, 4, 200, True), (2, 2, 513, True), (1, 1, 64, True), (3, 2, 1000, True),
                                          (1, 1, 8200, True), (1, 1, 8200, False)])
@pytest.mark.parametrize("checked", [0, 1])
def test_attention_bf16_stored_w64_kernel(vsa, B, H, T, masked, checked):
    """One wave per SIMD, both passes (optimistic first / every tile checked): every tile count modulo the ring and the
    loop unrolling, ragged last tiles (corrected in the epilogue / key bias), one-tile videos, arbitrary key masks with a
    fully masked tile and suffix padding, against fp64 on the stored operands and against the 8-wave kernel."""
    g = torch.Generator().manual_seed(1000 * T + B)
    q, k, v = (torch.randn(B, H, T, 64, generator=g) * 2.0 for _ in range(3))
    q16, k16, v16 = _stored_operands(q, k, v, (H * 64) ** -0.5)
    mask = None
    if masked:
        mask = torch.rand(B, T, generator=g) < 0.3
        mask[:, 0] = False
        if T > 130:
            mask[0, 64:128] = True
            mask[-1, T // 2:] = True
    ref = _attn_ref_stored(q16, k16, v16, mask)
    out = _run_attn_stored(vsa, q16, k16, v16, mask, 1, checked)
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() < BF16_STORED_REL * ref.abs().max().item()
    old = _run_attn_stored(vsa, q16, k16, v16, mask, 0)
    assert (out.double() - old.double()).abs().max().item() < 2 * BF16_STORED_REL * ref.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("T", [64, 200, 513, 1100])
@pytest.mark.parametrize("checked", [0, 1])
def test_attention_bf16_stored_w64_operand_layout_is_exact_on_a_permutation(vsa, T, checked):
    """Every index map of the one-wave-per-SIMD kernel (LDS-DMA pieces and their swizzles, both fragment reads, the
    accumulator-as-operand key order, both row blocks, the double-buffered V^T fragments, the epilogue), checked exactly:
    query i is a scaled copy of key pi(i), so its softmax row is one-hot to ~2^-30 and its output row must be V[pi(i)]
    (small integers: exact in bf16)."""
    B, H = 2, 2
    g = torch.Generator().manual_seed(T)
    k = (torch.randint(0, 2, (B, H, T, 64), generator=g) * 2 - 1).float()
    idx = torch.arange(T)
    k[..., :12] = ((idx[:, None] >> torch.arange(12)[None, :]) & 1).float() * 2 - 1
    perm = torch.stack([torch.randperm(T, generator=g) for _ in range(B * H)]).view(B, H, T)
    q = torch.gather(k, 2, perm[..., None].expand(-1, -1, -1, 64))
    v = torch.randint(-8, 9, (B, H, T, 64), generator=g).float()
    out = _run_attn_stored(vsa, (q * 16.0).to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16), None, 1, checked)
    want = torch.gather(v, 2, perm[..., None].expand(-1, -1, -1, 64)).permute(0, 2, 1, 3).reshape(B, T, H * 64)
    assert (out.float() - want).abs().max().item() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("T", [64, 127, 320, 1024])
def test_attention_bf16_stored_logits_beyond_2_to_the_14(vsa, T):
    """The 8-wave bf16 kernel carries its row constant through a bf16 operand, i.e. rounded to a bf16 grid point; it used to
    round UP, and beyond |c| = 2^14 (grid step 128) the row maximum's own P could drop under 2^-126, flush to zero and leave
    0 / 0 for the row (found by tools/fuzz_attn_w64.py on short videos, which take this kernel: q = 4, k = 5000 -> NaN while
    q = 100, k = 200 - the same logit, on a grid point - was fine).  Logits of 16 500 ... 30 000, on and off the grid, at
    lengths that take the 4-wave, the 8-wave and the one-wave-per-SIMD kernels: the softmax is one-hot, the row is V[pos]."""
    g = torch.Generator().manual_seed(T)
    for qa, kb in ((4.0, 5000.0), (2.0, 9000.0), (1.0, 16500.0), (1.0, 20000.0), (100.0, 200.0), (3.0, 9999.0), (1.0, 30000.0)):
        q = torch.randn(1, 2, T, 64, generator=g) * 0.1
        k = torch.randn(1, 2, T, 64, generator=g) * 0.1
        v = torch.randn(1, 2, T, 64, generator=g)
        pos = T // 2 + 3
        q[..., 0] = qa
        k[..., 0] = 0.0
        k[:, :, pos, 0] = kb
        q16, k16, v16 = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
        want = v16[:, :, pos].float().permute(0, 1, 2).reshape(1, 1, 2 * 64).expand(1, T, 128)
        for w64 in (0, 1):
            out = _run_attn_stored(vsa, q16, k16, v16, None, w64, 0)
            assert torch.isfinite(out).all(), (T, qa, kb, w64)
            assert (out.float() - want).abs().max().item() < 1e-6, (T, qa, kb, w64)


@pytest.mark.gpu
@pytest.mark.parametrize("checked", [0, 1])
def test_attention_bf16_stored_w64_every_gap_between_a_late_key_and_tile_0(vsa, checked):
    """The optimistic pass fixes a row's constant 60 below tile 0's maximum and tests its OUTPUTS afterwards.  A row whose
    late key sits g above tile 0's maximum sums l ~ 2^(60 + g): beyond 2^128 that is inf and is seen - but for g = 66, 67 the
    sum is finite while its reciprocal is a denormal, which v_rcp_f32 flushes to zero: every output of the row came out 0,
    nothing non-finite, no restart (found by tools/fuzz_attn_w64.py, round 4; the pass now also refuses l >= 2^120).  Here row
    i's late key is i / 4 above everything else: every gap from 0 to 250 in one call, both passes."""
    B, H, T = 1, 2, 1000
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(B, H, T, 64, generator=g) * 0.25 for _ in range(3))
    v = v * 4.0
    q[..., 0] = torch.arange(T, dtype=torch.float32) / 32.0        # q . k[pos] = (i / 32) * 8 = i / 4
    k[..., 0] = 0.0
    for pos in (700, 701):
        k[:, :, pos, :] = 0.0
        k[:, :, pos, 0] = 8.0
    q16, k16, v16 = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
    ref = _attn_ref_stored(q16, k16, v16, None)
    out = _run_attn_stored(vsa, q16, k16, v16, None, 1, checked)
    assert torch.isfinite(out).all()
    err = (out.double() - ref).abs().amax(dim=2)[0]
    worst = int(err.argmax())
    assert err.max().item() < BF16_STORED_REL * ref.abs().max().item(), "row %d (gap %.2f): %.3e" % (worst, worst / 4.0, err.max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("masked", [False, True])
def test_attention_bf16_stored_w64_rescue_and_raise(vsa, masked):
    """Scores of several hundred that rise late in the video: the optimistic pass overflows (its row constant is set once,
    from tile 0), finds the inf / NaN in its outputs and the block runs again in the checked form, whose rare path raises
    the constants; with a key mask the checked form runs alone.  Both against fp64 on the stored operands."""
    B, H, T = 2, 2, 700
    g = torch.Generator().manual_seed(99)
    q, k, v = (torch.randn(B, H, T, 64, generator=g) for _ in range(3))
    k[:, :, 300] = q.mean(dim=2) * 50.0 + 20.0
    k[:, :, 77] = -k[:, :, 300]
    k[:, :, 650] = q.mean(dim=2) * 120.0 + 40.0
    q16, k16, v16 = (q * 1.4426950408889634).to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
    mask = None
    if masked:
        mask = torch.rand(B, T, generator=g) < 0.2
        mask[:, 0] = False
    ref = _attn_ref_stored(q16, k16, v16, mask)
    for checked in (0, 1):
        out = _run_attn_stored(vsa, q16, k16, v16, mask, 1, checked)
        assert torch.isfinite(out).all()
        assert (out.double() - ref).abs().max().item() < BF16_STORED_REL * ref.abs().max().item()


@pytest.mark.gpu
def test_attention_bf16_stored_w64_is_deterministic_and_batch_invariant(vsa):
    """A video's rows depend on that video alone: the same video scored in a batch of three and alone gives the same bits
    (blocks never span videos, and the pass a block ends up in depends on its own rows only), run to run as well."""
    H, T = 4, 448
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(3, H, T, 64, generator=g) * 2.0 for _ in range(3))
    k[1, :, 400] = q[1].mean(dim=1) * 60.0 + 30.0         # video 1 takes the rescue path, videos 0 and 2 do not
    q16, k16, v16 = (q * 1.4426950408889634 * 0.25).to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
    full = _run_attn_stored(vsa, q16, k16, v16, None)
    again = _run_attn_stored(vsa, q16, k16, v16, None)
    assert torch.equal(full, again)
    for b in range(3):
        one = _run_attn_stored(vsa, q16[b:b + 1].contiguous(), k16[b:b + 1].contiguous(), v16[b:b + 1].contiguous(), None)
        assert torch.equal(one[0], full[b])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["M-A", "M-B", "M-B512"])
def test_bf16_attention_mode_end_to_end(vsa, cfg):
    """SimNet.attention_dtype = 'bf16' against the fp32 oracle, ragged batch; head dim 64, 32 and 128 (BASELINE's M-B:
    4 heads, d_model 512 - bf16 attention with exact Linear kernels)."""
    synth = vsa.synth
    d, H, L = {"M-A": (256, 4, 4), "M-B": (256, 8, 6), "M-B512": (512, 4, 3)}[cfg]
    sd = synth.make_state_dict(d, L, 71, trained_like=True)
    lengths = [400, 333, 64, 1]
    x = synth.make_features(4, 400, 72, "pool5", lengths=lengths)
    mask = synth.padding_mask(x)
    m = vsa.SimNet(num_heads=H, d_model=d, num_layers=L, sparsity=0.0, dropout=0.3)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval()
    m.attention_dtype = "bf16"
    with torch.no_grad():
        logits, hidden = m(x.to(_dev()), mask.to(_dev()))
        exact = vsa.SimNet(num_heads=H, d_model=d, num_layers=L, sparsity=0.0, dropout=0.3)
        exact.load_state_dict(sd, strict=True)
        exact = exact.to(_dev()).eval()
        l32, _ = exact(x.to(_dev()), mask.to(_dev()))
        rl, rh = oracle_forward(sd, x, mask, H)
    valid = ~mask
    err = (logits.cpu() - rl).abs().squeeze(-1)[valid].max().item()
    serr = (torch.sigmoid(logits.cpu()) - torch.sigmoid(rl)).abs().squeeze(-1)[valid].max().item()
    print("bf16 attention %s: logit err %.3e score err %.3e (fp32 path %.3e)" % (
        cfg, err, serr, (l32.cpu() - rl).abs().squeeze(-1)[valid].max().item()))
    assert err < BF16_LOGIT_TOL and serr < BF16_SCORE_TOL
    assert (logits.cpu() - l32.cpu()).abs().max().item() > 0        # the flag really switches kernels
    with pytest.raises(ValueError):
        m.attention_dtype = "fp16"
    assert m.attention_dtype == "bf16"


@pytest.mark.gpu
def test_bf16_attention_long_video(vsa):
    """BASELINE configs[4] as named: T=8192, 2048-d features, bf16 attention (oracle: fp32 restatement)."""
    synth = vsa.synth
    sd = synth.make_state_dict(256, 2, 61, in_features=2048, max_len=8192)
    x = synth.make_features(1, 8192, 62, "randn", in_features=2048)
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=2, sparsity=0.0, dropout=0.3, in_features=2048, pe_len=8192)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval()
    m.attention_dtype = "bf16"
    with torch.no_grad():
        logits, hidden = m(x.to(_dev()))
        torch.set_num_threads(16)
        rl, rh = oracle_forward(sd, x, None, 4)
    err = (logits.cpu() - rl).abs().max().item()
    print("bf16 attention T=8192: logit err %.3e" % err)
    assert err < BF16_LOGIT_TOL





# ---- opt-in bf16 Linear kernels (VS_FLAG_BF16_LINEAR) ---------------------------------------------------
# The checker multiplies the SAME bf16-rounded operands in fp64, so what is left is fp32 accumulation order:
# 1e-4 absolute on O(1) outputs.  Against unrounded operands the error is the bf16 rounding itself.
def _rb(t):
    return t.to(torch.bfloat16).double()


@pytest.mark.parametrize("M,N,K,relu,T", [(300, 256, 1024, 0, 0), (129, 1024, 256, 1, 0), (64, 768, 256, 0, 0),
                                          (1000, 256, 1024, 0, 250), (37, 512, 2048, 0, 37), (2048, 2048, 512, 1, 0),
                                          (1, 32, 32, 0, 0)])
def test_linear_bf16_kernel(vsa, M, N, K, relu, T):
    lib = vsa._lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    pe = torch.randn(T, N, generator=g) if T else None
    ref = _rb(A) @ _rb(W).t() + b.double()
    if relu:
        ref = F.relu(ref)
    if T:
        ref = ref + pe.double().repeat(M // T, 1)
    dA, dW, db = A.to(_dev()), W.to(_dev()), b.to(_dev())
    dpe = pe.to(_dev()) if T else None
    out = torch.full((M, N), float("nan"), device=_dev())
    vsa._lib.check(lib.vs_linear_bf16(dA.data_ptr(), dW.data_ptr(), db.data_ptr(), out.data_ptr(), M, N, K, relu,
                                      dpe.data_ptr() if T else None, T, _stream()))
    torch.cuda.synchronize()
    assert (out.cpu().double() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("M,N,K,relu,c16", [(300, 256, 1024, 0, 0), (129, 1024, 256, 1, 1), (64, 768, 256, 0, 0), (1000, 512, 2048, 0, 0),
                                            (2048, 2048, 512, 1, 1), (1, 32, 32, 0, 0), (257, 96, 64, 1, 0), (4096, 1536, 512, 0, 1),
                                            (130, 288, 96, 0, 1)])
def test_linear_bf16_operands_kernel(vsa, M, N, K, relu, c16):
    """The wide models' bf16 Linear (vs_gemm_ring.hip): bf16 operands from device memory by LDS-DMA, fp32 accumulation.
    vs_to_bf16 must round exactly like torch (nearest even); the product against float64 on the SAME rounded operands
    leaves fp32 accumulation order only (1e-4 on O(1) outputs); a bf16 output is the nearest bf16 of the fp32 result up
    to that accumulation noise (one bf16 ulp)."""
    lib = vsa._lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    ref = _rb(A) @ _rb(W).t() + b.double()
    if relu:
        ref = F.relu(ref)
    dA, dW, db = A.to(_dev()), W.to(_dev()), b.to(_dev())
    pad = (-(M * K)) % 8                                  # vs_to_bf16 converts 8 elements per thread
    A16 = torch.empty(M * K + pad, dtype=torch.bfloat16, device=_dev())
    W16 = torch.empty(N * K, dtype=torch.bfloat16, device=_dev())
    srcA = torch.cat([dA.reshape(-1), torch.zeros(pad, device=_dev())]) if pad else dA.reshape(-1)
    vsa._lib.check(lib.vs_to_bf16(srcA.data_ptr(), A16.data_ptr(), M * K + pad, _stream()))
    vsa._lib.check(lib.vs_to_bf16(dW.data_ptr(), W16.data_ptr(), N * K, _stream()))
    torch.cuda.synchronize()
    assert torch.equal(A16[:M * K].cpu(), A.reshape(-1).to(torch.bfloat16)) and torch.equal(W16.cpu(), W.reshape(-1).to(torch.bfloat16))
    out = torch.full((M, N), float("nan"), device=_dev(), dtype=torch.bfloat16 if c16 else torch.float32)
    vsa._lib.check(lib.vs_linear_bf16_operands(A16.data_ptr(), W16.data_ptr(), db.data_ptr(), out.data_ptr(), M, N, K, relu, c16, _stream()))
    torch.cuda.synchronize()
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    if c16:
        assert ((got - ref).abs() <= 2.0 ** -8 * ref.abs() + 1e-4).all()
    else:
        assert (got - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("M,N,K,res", [(400, 256, 1024, 0), (129, 256, 1024, 1), (300, 1024, 256, 0), (2048, 512, 2048, 1), (64, 768, 256, 0),
                                       (1, 32, 32, 0), (1000, 256, 96, 1)])
def test_linear_bf16_stored_a_kernel(vsa, M, N, K, res):
    """gemm_nt_128's bf16 form with A read as bf16 from memory (the bf16 training mode's fc2 and fc1 input gradient, the latter
    with the residual gradient [M,N] added in the epilogue): same result as vs_linear_bf16 on the fp32 tensor that rounds to
    this bf16 - BIT for bit (the fp32-operand form rounds to the same values on its way into LDS)."""
    lib = vsa._lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g) if res else None
    ref = A.double() @ _rb(W).t() + b.double() + (R.double() if res else 0.0)
    dA16, dA32, dW, db = A.to(_dev()), A.float().to(_dev()), W.to(_dev()), b.to(_dev())
    dR = R.to(_dev()) if res else None
    out = torch.full((M, N), float("nan"), device=_dev())
    vsa._lib.check(lib.vs_linear_bf16_a16(dA16.data_ptr(), dW.data_ptr(), db.data_ptr(), out.data_ptr(), M, N, K, dR.data_ptr() if res else None, _stream()))
    want = torch.full((M, N), float("nan"), device=_dev())
    vsa._lib.check(lib.vs_linear_bf16(dA32.data_ptr(), dW.data_ptr(), db.data_ptr(), want.data_ptr(), M, N, K, 0, dR.data_ptr() if res else None, M if res else 0, _stream()))
    torch.cuda.synchronize()
    assert (out.cpu().double() - ref).abs().max().item() < 1e-4
    assert torch.equal(out, want)


@pytest.mark.parametrize("B,T,d,H,c16", [(2, 200, 512, 4, 0), (3, 130, 768, 12, 1), (1, 77, 1024, 8, 0), (2, 64, 1024, 16, 1)])
def test_qkv_projection_bf16_operands_kernel(vsa, B, T, d, H, c16):
    """The same kernel's q/k/v epilogue: [3][B][H][T][dh], q pre-multiplied by scale * log2(e) in the bf16-output form."""
    lib = vsa._lib.load()
    g = torch.Generator().manual_seed(B * T + d)
    h = torch.randn(B * T, d, generator=g)
    W = torch.randn(3 * d, d, generator=g) / math.sqrt(d)
    b = torch.randn(3 * d, generator=g)
    ref = (_rb(h) @ _rb(W).t() + b.double()).view(B, T, 3, H, d // H).permute(2, 0, 3, 1, 4).contiguous()
    if c16:
        ref[0] *= d ** -0.5 * 1.4426950408889634
    h16 = h.to(torch.bfloat16).to(_dev())
    W16 = W.to(torch.bfloat16).to(_dev())
    out = torch.full((3, B, H, T, d // H), float("nan"), device=_dev(), dtype=torch.bfloat16 if c16 else torch.float32)
    vsa._lib.check(lib.vs_qkv_proj_bf16_operands(h16.data_ptr(), W16.data_ptr(), b.to(_dev()).data_ptr(), out.data_ptr(), B, T, d, H, c16, _stream()))
    torch.cuda.synchronize()
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    if c16:
        assert ((got - ref).abs() <= 2.0 ** -8 * ref.abs() + 1e-4).all()
    else:
        assert (got - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("M,N,K,nc,sig", [(300, 256, 256, 0, 0), (100, 256, 1024, 1, 0), (64, 128, 512, 3, 0),
                                          (1000, 192, 320, 2, 1), (33, 64, 256, 1, 0)])
def test_linear_residual_layernorm_bf16_kernel(vsa, M, N, K, nc, sig):
    lib = vsa._lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    gam, bet = 1 + 0.1 * torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g)
    sw, sb = torch.randn(max(nc, 1), N, generator=g) / math.sqrt(N), torch.randn(max(nc, 1), generator=g)
    y = F.layer_norm(_rb(A) @ _rb(W).t() + b.double() + res.double(), (N,), gam.double(), bet.double(), 1e-5)
    sc = F.linear(y, sw.double(), sb.double())
    if sig:
        sc = torch.sigmoid(sc)
    d = [t.to(_dev()) for t in (A, W, b, res, gam, bet, sw, sb)]
    out = torch.full((M, N), float("nan"), device=_dev())
    scores = torch.full((M, max(nc, 1)), float("nan"), device=_dev())
    vsa._lib.check(lib.vs_linear_residual_layernorm_bf16(
        d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), d[5].data_ptr(),
        out.data_ptr(), M, N, K, d[6].data_ptr() if nc else None, d[7].data_ptr() if nc else None, nc, sig,
        scores.data_ptr() if nc else None, _stream()))
    torch.cuda.synchronize()
    assert (out.cpu().double() - y).abs().max().item() < 1e-4
    if nc:
        assert (scores.cpu().double() - sc).abs().max().item() < 1e-4


BF16_FULL_LOGIT_TOL = tol.BF16_LOGIT_TOL   # all products on the bf16 pipe (measured on trained-like weights: 4.3e-3 logits, 1.0e-3 scores)
BF16_FULL_SCORE_TOL = tol.BF16_SCORE_TOL


@pytest.mark.parametrize("cfg", ["M-A", "M-B8"])
def test_bf16_compute_mode_end_to_end(vsa, lp_linear_everywhere, cfg):
    """SimNet.set_compute_dtype('bf16'): every matrix product on the bf16 pipe, against the fp32 oracle."""
    synth = vsa.synth
    d, H, L = (256, 4, 4) if cfg == "M-A" else (256, 8, 6)
    sd = synth.make_state_dict(d, L, 71, trained_like=True)
    lengths = [400, 333, 64, 1]
    x = synth.make_features(4, 400, 72, "pool5", lengths=lengths)
    mask = synth.padding_mask(x)
    m = vsa.SimNet(num_heads=H, d_model=d, num_layers=L, sparsity=0.0, dropout=0.3)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval().set_compute_dtype("bf16")
    with torch.no_grad():
        logits, hidden = m(x.to(_dev()), mask.to(_dev()))
        s = m.score(x.to(_dev()), mask.to(_dev()))
        rl, rh = oracle_forward(sd, x, mask, H)
    valid = ~mask
    err = (logits.cpu() - rl).abs().squeeze(-1)[valid].max().item()
    serr = (s.cpu() - torch.sigmoid(rl).squeeze(-1)).abs()[valid].max().item()
    herr = (hidden.cpu() - rh).abs()[valid].max().item()
    print("bf16 compute %s: logit err %.3e score err %.3e hidden err %.3e" % (cfg, err, serr, herr))
    assert err < BF16_FULL_LOGIT_TOL and serr < BF16_FULL_SCORE_TOL
    m.set_compute_dtype("fp32")
    with torch.no_grad():
        l32, _ = m(x.to(_dev()), mask.to(_dev()))
    assert (l32.cpu() - rl).abs().squeeze(-1)[valid].max().item() < TOL     # and back: the exact path is untouched


def test_bf16_compute_long_video(vsa, lp_linear_everywhere):
    """BASELINE configs[4]: T=8192, 2048-d features, all products bf16 (oracle: fp32 restatement)."""
    synth = vsa.synth
    sd = synth.make_state_dict(256, 2, 61, in_features=2048, max_len=8192)
    x = synth.make_features(1, 8192, 62, "randn", in_features=2048)
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=2, sparsity=0.0, dropout=0.3, in_features=2048, pe_len=8192)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval().set_compute_dtype("bf16")
    with torch.no_grad():
        logits, hidden = m(x.to(_dev()))
        torch.set_num_threads(16)
        rl, rh = oracle_forward(sd, x, None, 4)
    err = (logits.cpu() - rl).abs().max().item()
    print("bf16 compute T=8192: logit err %.3e" % err)
    assert err < BF16_FULL_LOGIT_TOL


@pytest.mark.parametrize("cfg", ["M-A", "M-B8", "M-A-short"])
def test_bf16_storage_is_bit_identical_to_fp32_storage(vsa, lp_linear_everywhere, cfg):
    """In the bf16 mode q*scale, k, v, the attention output and the MLP hidden tensor are written to HBM as bf16 by
    their producers (every consumer rounds them to bf16 on entry anyway).  VS_LP_STORE32=1 keeps them fp32: the two
    must agree BIT FOR BIT - padded batch, masked batch, packed ragged batch, 4- and 8-wave attention blocks, head dim
    64 and 32, the score head and the hidden state."""
    synth = vsa.synth
    d, H, L = (256, 8, 3) if cfg == "M-B8" else (256, 4, 2)
    sd = synth.make_state_dict(d, L, 171, trained_like=True)
    lengths = [130, 77, 64, 1] if cfg == "M-A-short" else [512, 333, 256, 31]
    x = synth.make_features(4, max(lengths), 172, "pool5", lengths=lengths)
    mask = synth.padding_mask(x)
    m = vsa.SimNet(num_heads=H, d_model=d, num_layers=L, sparsity=0.0, dropout=0.3)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval().set_compute_dtype("bf16")
    dx, dm = x.to(_dev()), mask.to(_dev())
    packed = torch.cat([x[i, :t] for i, t in enumerate(lengths)]).to(_dev())
    lens = torch.tensor(lengths, dtype=torch.int32)

    def run():
        with torch.no_grad():
            a = m(dx, dm)
            b = m(dx[:1].contiguous())
            c = m.score_packed(packed, lens)
        return [t.clone() for t in (*a, *b, c)]

    try:
        vsa._lib.set_option("VS_LP_MLP_UNFUSED", 1)      # (the fused MLP / layer-tail kernels have their own tests below)
        # the storage forms of ONE attention kernel are compared: the bf16-stored head-dim-64 form otherwise runs on the
        # one-wave-per-SIMD kernel (other summation order; its own tests: test_attention_bf16_stored_*)
        vsa._lib.set_option("VS_ATTN_W64", 0)
        vsa._lib.set_option("VS_LP_STORE32", 1)
        ref = run()
        vsa._lib.set_option("VS_LP_STORE32", -1)
        got = run()
    finally:
        vsa._lib.set_option("VS_LP_STORE32", -1)
        vsa._lib.set_option("VS_LP_MLP_UNFUSED", -1)
        vsa._lib.set_option("VS_ATTN_W64", -1)
    for g, r in zip(got, ref):
        assert torch.isfinite(g).all() and torch.equal(g, r)


def _bf16_round(t):
    return t.to(torch.bfloat16).double()


@pytest.mark.parametrize("M,nc,sig", [(256, 0, 0), (1000, 1, 1), (257, 3, 0), (31, 1, 0), (4096, 1, 0), (70000, 1, 0)])
def test_mlp_block_bf16_kernel(vsa, M, nc, sig):
    """vs_mlp_block_bf16 (fc1 + ReLU + fc2 + residual + LayerNorm + score head as one kernel, bf16 matrix pipe) against
    a float64 evaluation that shares its rounding points: h, W1, W2 and relu(fc1) rounded to bf16, everything else
    exact.  The kernel accumulates in fp32, so a hidden activation within fp32 rounding of a bf16 tie can round the
    other way (one bf16 ulp of one of 1024 terms): bar 2e-3 on the LayerNorm output (measured: ~1e-5 typical)."""
    lib = vsa._lib.load()
    sd = vsa.synth.make_state_dict(256, 2, 300 + M, trained_like=True, num_classes=max(nc, 1))
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=2, sparsity=0.0, dropout=0.3, num_classes=max(nc, 1))
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval()
    packed = m._packed_weights(_dev())
    g = torch.Generator().manual_seed(M)
    h = torch.randn(M, 256, generator=g)
    pre = "encoder.module_list.1."
    W1, b1 = sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"].double()
    W2, b2 = sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"].double()
    act = _bf16_round(F.relu(_bf16_round(h) @ _bf16_round(W1).T + b1).float())
    y = act @ _bf16_round(W2).T + b2 + h.double()
    ref = F.layer_norm(y, (256,), sd[pre + "norm2.weight"].double(), sd[pre + "norm2.bias"].double(), 1e-5)
    sc = ref @ sd["final_layer.weight"].double().T + sd["final_layer.bias"].double()
    if sig:
        sc = torch.sigmoid(sc)
    dh = h.to(_dev())
    out = torch.full((M, 256), float("nan"), device=_dev())
    scores = torch.full((M, max(nc, 1)), float("nan"), device=_dev())
    vsa._lib.check(lib.vs_mlp_block_bf16(packed.handle, 1, dh.data_ptr(), out.data_ptr(), M, 1 if nc else 0, sig,
                                         scores.data_ptr(), _stream()))
    torch.cuda.synchronize()
    err = (out.cpu().double() - ref).abs()
    print("mlp_block_bf16 M=%d: max err %.2e, median %.2e" % (M, err.max().item(), err.median().item()))
    assert err.max().item() < 2e-3 and err.median().item() < 1e-5
    if nc:
        assert (scores.cpu().double() - sc).abs().max().item() < 2e-3


@pytest.mark.parametrize("tile256", [0, 1])
@pytest.mark.parametrize("cfg", ["M-A", "M-B8", "M-A-ragged"])
def test_bf16_layer_tail_kernel_is_bit_identical_to_outproj_then_mlp_kernel(vsa, lp_linear_everywhere, cfg, tile256):
    """With the attention output stored as bf16, the bf16 mode runs out-projection + residual + norm1 + MLP block + norm2
    (+ score head) as ONE kernel (vs_mlp_fused.hip, TAIL), which then also projects its rows to the NEXT layer's q/k/v
    (QKV epilogue); the embedding runs on the same design with the first layer's QKV behind it (embed_qkv_bf16).  All
    three multiply in the same order as the stand-alone kernels they replace (gemm_ln_rows; gemm_nt_128<EPI_QKV, C16>;
    gemm_nt_128<EPI_PE>) and share their epilogue arithmetic, so against VS_LP_TAIL_UNFUSED=1 / VS_LP_QKV_UNFUSED=1 / VS_LP_EMBED_UNFUSED=1
    (those kernels, h1 and the layer output re-read from HBM) logits, scores and hidden state must agree BIT FOR BIT -
    padded, masked and packed batches, head dim 64 and 32, in-place (middle layers) and out-of-place (last layer)."""
    synth = vsa.synth
    d, H, L = (256, 8, 3) if cfg == "M-B8" else (256, 4, 3)
    sd = synth.make_state_dict(d, L, 371, trained_like=True)
    lengths = [257, 1, 100, 513, 7] if cfg == "M-A-ragged" else [512, 333, 256, 31]
    x = synth.make_features(len(lengths), max(lengths), 372, "pool5", lengths=lengths)
    mask = synth.padding_mask(x)
    m = vsa.SimNet(num_heads=H, d_model=d, num_layers=L, sparsity=0.0, dropout=0.3)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval().set_compute_dtype("bf16")
    dx, dm = x.to(_dev()), mask.to(_dev())
    packed = torch.cat([x[i, :t] for i, t in enumerate(lengths)]).to(_dev())
    lens = torch.tensor(lengths, dtype=torch.int32)

    def run():
        with torch.no_grad():
            a = m(dx, dm)
            b = m.score(dx, dm)
            c = m.score_packed(packed, lens)
        return [t.clone() for t in (*a, b, c)]

    try:
        # both block shapes of the fused kernels: 4 waves x 32 rows (the default at these row counts) and 8 waves (pinned)
        vsa._lib.set_option("VS_LP_TILE256", tile256)
        vsa._lib.set_option("VS_LP_TAIL_UNFUSED", 1)
        vsa._lib.set_option("VS_LP_QKV_UNFUSED", 1)
        ref = run()                                      # out-projection kernel, MLP kernel, QKV kernel per layer
        vsa._lib.set_option("VS_LP_TAIL_UNFUSED", -1)
        mid = run()                                      # layer-tail kernel, QKV kernel per layer
        vsa._lib.set_option("VS_LP_QKV_UNFUSED", -1)
        vsa._lib.set_option("VS_LP_TAIL_UNFUSED", 1)
        mid2 = run()                                     # out-projection kernel, MLP kernel with the next layer's QKV behind it
        vsa._lib.set_option("VS_LP_TAIL_UNFUSED", -1)
        vsa._lib.set_option("VS_LP_EMBED_UNFUSED", 1)
        mid3 = run()                                     # generic embedding GEMM + first QKV kernel, then layer-tail kernels
        vsa._lib.set_option("VS_LP_EMBED_UNFUSED", -1)
        got = run()                                      # all fused: embedding + QKV kernel, layer-tail kernels with the next QKV
    finally:
        vsa._lib.set_option("VS_LP_TAIL_UNFUSED", -1)
        vsa._lib.set_option("VS_LP_QKV_UNFUSED", -1)
        vsa._lib.set_option("VS_LP_EMBED_UNFUSED", -1)
        vsa._lib.set_option("VS_LP_TILE256", -1)
    for g, a, b, c, r in zip(got, mid, mid2, mid3, ref):
        assert torch.isfinite(g).all() and torch.equal(a, r) and torch.equal(b, r) and torch.equal(c, r) and torch.equal(g, r)


@pytest.mark.parametrize("cfg", ["M-A", "M-A-ragged"])
def test_bf16_fused_mlp_matches_the_two_kernel_path(vsa, lp_linear_everywhere, cfg):
    """bf16 mode, d_model 256: fc1 + ReLU + fc2 + residual + LayerNorm (+ score head) run as ONE kernel whose hidden
    activations stay in registers (vs_mlp_fused.hip).  Same rounding points as the two-kernel path (VS_LP_MLP_UNFUSED=1);
    only the order of the 16 products inside an MFMA step differs, so the two agree to fp32 rounding amplified by an
    occasional 1-ulp flip of a bf16-rounded activation - and every later layer re-rounds its inputs to bf16, so a 1e-5
    difference after layer 1 flips some 2^-9-relative roundings in layer 2.  Bar 1e-2 on logits and hidden state
    (measured 1e-3 / 3e-3 over three layers; a wrong index map would show as O(1)); the kernel's own arithmetic is
    pinned at 2e-3 with shared rounding points by test_mlp_block_bf16_kernel."""
    synth = vsa.synth
    sd = synth.make_state_dict(256, 3, 271, trained_like=True)
    lengths = [700, 333, 256, 31] if cfg == "M-A" else [257, 1, 100, 513, 7]
    x = synth.make_features(len(lengths), max(lengths), 272, "pool5", lengths=lengths)
    mask = synth.padding_mask(x)
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=3, sparsity=0.0, dropout=0.3)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval().set_compute_dtype("bf16")
    dx, dm = x.to(_dev()), mask.to(_dev())

    def run():
        with torch.no_grad():
            a = m(dx, dm)
            b = m.score(dx, dm)
        return [t.clone() for t in (*a, b)]

    try:
        vsa._lib.set_option("VS_LP_MLP_UNFUSED", 1)
        ref = run()
        vsa._lib.set_option("VS_LP_MLP_UNFUSED", -1)
        got = run()
    finally:
        vsa._lib.set_option("VS_LP_MLP_UNFUSED", -1)
    valid = ~dm
    for g, r in zip(got, ref):
        assert torch.isfinite(g).all()
        gv, rv = (g[valid], r[valid]) if g.dim() >= 2 and g.shape[:2] == valid.shape else (g, r)
        err = (gv - rv).abs().max().item()
        print("fused vs two-kernel MLP (%s): %.3e" % (cfg, err))
        assert err < 1e-2
    rl, rh = oracle_forward(sd, x, mask, 4)
    assert (got[0].cpu() - rl).abs().squeeze(-1)[~mask].max().item() < BF16_FULL_LOGIT_TOL


# ---- opt-in fp32 emulation on the f16 matrix pipe (VS_FLAG_F16X3_LINEAR, "fp16x3") ------------------------
# Operands are split into two f16 halves (22 bits) and three products are accumulated in fp32, so the checker is
# the plain fp64 product of the UNROUNDED operands, and the bar is the fp32 path's own 1e-4 (per-kernel 5e-5).
@pytest.mark.parametrize("M,N,K,relu,T", [(300, 256, 1024, 0, 0), (129, 1024, 256, 1, 0), (64, 768, 256, 0, 0),
                                          (1000, 256, 1024, 0, 250), (37, 512, 2048, 0, 37), (2048, 2048, 512, 1, 0),
                                          (1, 32, 32, 0, 0)])
def test_linear_f16x3_kernel(vsa, M, N, K, relu, T):
    lib = vsa._lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    pe = torch.randn(T, N, generator=g) if T else None
    ref = F.linear(A.double(), W.double(), b.double())
    if relu:
        ref = F.relu(ref)
    if T:
        ref = ref + pe.double().repeat(M // T, 1)
    dA, dW, db = A.to(_dev()), W.to(_dev()), b.to(_dev())
    dpe = pe.to(_dev()) if T else None
    out = torch.full((M, N), float("nan"), device=_dev())
    vsa._lib.check(lib.vs_linear_f16x3(dA.data_ptr(), dW.data_ptr(), db.data_ptr(), out.data_ptr(), M, N, K, relu,
                                       dpe.data_ptr() if T else None, T, _stream()))
    torch.cuda.synchronize()
    err = (out.cpu().double() - ref).abs().max().item()
    print("f16x3 linear M=%d N=%d K=%d: err %.2e" % (M, N, K, err))
    assert err < 5e-5


@pytest.mark.parametrize("M,N,K,nc,sig", [(300, 256, 256, 0, 0), (100, 256, 1024, 1, 0), (64, 128, 512, 3, 0),
                                          (1000, 192, 320, 2, 1), (33, 64, 256, 1, 0)])
def test_linear_residual_layernorm_f16x3_kernel(vsa, M, N, K, nc, sig):
    lib = vsa._lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    gam, bet = 1 + 0.1 * torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g)
    sw, sb = torch.randn(max(nc, 1), N, generator=g) / math.sqrt(N), torch.randn(max(nc, 1), generator=g)
    y = F.layer_norm(F.linear(A.double(), W.double(), b.double()) + res.double(), (N,), gam.double(), bet.double(), 1e-5)
    sc = F.linear(y, sw.double(), sb.double())
    if sig:
        sc = torch.sigmoid(sc)
    d = [t.to(_dev()) for t in (A, W, b, res, gam, bet, sw, sb)]
    out = torch.full((M, N), float("nan"), device=_dev())
    scores = torch.full((M, max(nc, 1)), float("nan"), device=_dev())
    vsa._lib.check(lib.vs_linear_residual_layernorm_f16x3(
        d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), d[5].data_ptr(),
        out.data_ptr(), M, N, K, d[6].data_ptr() if nc else None, d[7].data_ptr() if nc else None, nc, sig,
        scores.data_ptr() if nc else None, _stream()))
    torch.cuda.synchronize()
    assert (out.cpu().double() - y).abs().max().item() < 5e-5
    if nc:
        assert (scores.cpu().double() - sc).abs().max().item() < 5e-5


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_f16x3_linear_mode_matches_reference_golden(vsa, lp_linear_everywhere, case):
    """linear_dtype='fp16x3' against the reference-generated goldens, at the fp32 path's own 1e-4 bar."""
    g = load_golden(case["name"])
    sd, x, mask = build_case(vsa.synth, case)
    m = _model(vsa, case, sd)
    m.linear_dtype = "fp16x3"
    with torch.no_grad():
        logits, hidden = m(x.to(_dev()), None if mask is None else mask.to(_dev()))
    valid = torch.ones(x.shape[:2], dtype=torch.bool) if mask is None else ~mask
    dl = (logits.cpu() - g["logits"])[valid].abs().max().item()
    dh = (hidden.cpu()[:, g["rows"]] - g["hidden"])[valid[:, g["rows"]]].abs().max().item()
    print("f16x3 linear %s: logits %.2e hidden %.2e" % (case["name"], dl, dh))
    assert dl < TOL and dh < TOL, (dl, dh)


def _run_attn_f16x3(vsa, q, k, v, mask, scale):
    lib = vsa._lib.load()
    B, H, T, dh = q.shape
    dq, dk, dv = q.to(_dev()), k.to(_dev()), v.to(_dev())
    dm = mask.to(_dev()) if mask is not None else None
    out = torch.full((B, T, H * dh), float("nan"), device=_dev())
    vsa._lib.check(lib.vs_attention_f16x3(dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                          dm.data_ptr() if dm is not None else None, out.data_ptr(), B, H, T, dh,
                                          scale, _stream()))
    torch.cuda.synchronize()
    return out.cpu()


@pytest.mark.parametrize("B,H,T,dh,masked", [(2, 4, 320, 64, False), (1, 4, 1024, 64, False), (2, 4, 200, 64, True),
                                             (1, 8, 65, 32, True), (1, 4, 31, 64, False), (3, 8, 257, 32, False),
                                             (1, 1, 1, 64, False)])
def test_attention_f16x3_kernel(vsa, B, H, T, dh, masked):
    """fp32 attention emulated on the f16 pipe, against the fp64 reference at the fp32 kernel's own tolerance."""
    g = torch.Generator().manual_seed(T + dh)
    q, k, v = (torch.randn(B, H, T, dh, generator=g) * 2.0 for _ in range(3))
    mask = vsa.synth.random_mask(B, T, 3) if masked else None
    scale = (H * dh) ** -0.5
    ref = _attn_ref(q, k, v, mask, scale)
    out = _run_attn_f16x3(vsa, q, k, v, mask, scale)
    err = (out.double() - ref).abs().max().item()
    print("f16x3 attention T=%d dh=%d: err %.2e" % (T, dh, err))
    assert err < 2e-5


@pytest.mark.parametrize("T,dh", [(64, 64), (200, 64), (513, 64), (300, 32)])
def test_attention_f16x3_operand_layout_is_exact_on_a_permutation(vsa, T, dh):
    B, H = 2, 2
    g = torch.Generator().manual_seed(T)
    k = (torch.randint(0, 2, (B, H, T, dh), generator=g) * 2 - 1).float()
    idx = torch.arange(T)
    k[..., :12] = ((idx[:, None] >> torch.arange(12)[None, :]) & 1).float() * 2 - 1
    perm = torch.stack([torch.randperm(T, generator=g) for _ in range(B * H)]).view(B, H, T)
    q = torch.gather(k, 2, perm[..., None].expand(-1, -1, -1, dh))
    v = torch.randint(-8, 9, (B, H, T, dh), generator=g).float() + 0.001        # needs the lo half too
    out = _run_attn_f16x3(vsa, q * 16.0, k, v, None, 1.0)
    want = torch.gather(v, 2, perm[..., None].expand(-1, -1, -1, dh)).permute(0, 2, 1, 3).reshape(B, T, H * dh)
    assert (out - want).abs().max().item() < 2e-6


def test_attention_f16x3_rescale_branch(vsa):
    B, H, T, dh = 1, 4, 512, 64
    g = torch.Generator().manual_seed(99)
    q, k, v = (torch.randn(B, H, T, dh, generator=g) for _ in range(3))
    k[:, :, 300] = q.mean(dim=2) * 50.0 + 20.0
    k[:, :, 77] = -k[:, :, 300]
    ref = _attn_ref(q, k, v, None, 1.0)
    out = _run_attn_f16x3(vsa, q, k, v, None, 1.0)
    err = (out.double() - ref).abs().max().item()
    print("f16x3 attention rescale: err %.2e" % err)
    assert err < 5e-4          # scores of several hundred: 2^-22 of them is ~1e-4 in the exponent


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_f16x3_compute_mode_matches_reference_golden(vsa, case, kernel_path):
    """set_compute_dtype('fp16x3') (every product emulated on the f16 pipe) against the reference-generated
    goldens, at the fp32 path's own 1e-4 bar - through its latency kernels ("auto") and its tiled kernels."""
    # wide models (d_model 512, head dim 128): the plain projections are emulated, the rest stays exact
    g = load_golden(case["name"])
    sd, x, mask = build_case(vsa.synth, case)
    m = _model(vsa, case, sd).set_compute_dtype("fp16x3")
    with torch.no_grad():
        logits, hidden = m(x.to(_dev()), None if mask is None else mask.to(_dev()))
    valid = torch.ones(x.shape[:2], dtype=torch.bool) if mask is None else ~mask
    dl = (logits.cpu() - g["logits"])[valid].abs().max().item()
    dh = (hidden.cpu()[:, g["rows"]] - g["hidden"])[valid[:, g["rows"]]].abs().max().item()
    print("f16x3 %s: logits %.2e hidden %.2e" % (case["name"], dl, dh))
    assert dl < TOL and dh < TOL, (dl, dh)


def test_f16x3_full_size_batch_against_exact_path(vsa):
    """BASELINE configs[2] size (B=64, T=1024): the emulated path against the exact fp32 MFMA path."""
    synth = vsa.synth
    sd = synth.make_state_dict(256, 4, 1234)
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=4, sparsity=0.0, dropout=0.3)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval()
    x = torch.randn(64, 1024, 1024, generator=torch.Generator().manual_seed(5)).to(_dev())
    with torch.no_grad():
        l32, h32 = m(x)
        m.set_compute_dtype("fp16x3")
        l16, h16 = m(x)
    dl, dh = (l16 - l32).abs().max().item(), (h16 - h32).abs().max().item()
    print("f16x3 vs exact at B=64 T=1024: logits %.2e hidden %.2e" % (dl, dh))
    assert dl < TOL and dh < TOL
    # the emulated latency kernels (one video alone) multiply in the same order as the emulated tiled kernels
    # (the video inside the batch of 64): bit-identical scores whatever the batch
    with torch.no_grad():
        l1, h1 = m(x[5:6])
    assert torch.equal(l1[0], l16[5]) and torch.equal(h1[0], h16[5])


def test_f16x3_ragged_padded_batch_matches_oracle(vsa, lp_linear_everywhere):
    """fp16x3 on a right-padded batch large enough for the emulated Linear kernels (1600 rows): the 1000.0 padding
    sentinel (collate_fn_train) must survive the f16 split, valid frames must meet the 1e-4 bar."""
    synth = vsa.synth
    sd = synth.make_state_dict(256, 4, 71, trained_like=True)
    lengths = [400, 333, 64, 1]
    x = synth.make_features(4, 400, 72, "pool5", lengths=lengths)
    mask = synth.padding_mask(x)
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=4, sparsity=0.0, dropout=0.3)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval().set_compute_dtype("fp16x3")
    with torch.no_grad():
        logits, hidden = m(x.to(_dev()), mask.to(_dev()))
        rl, rh = oracle_forward(sd, x, mask, 4)
    valid = ~mask
    assert torch.isfinite(logits.cpu()[valid]).all()
    dl = (logits.cpu() - rl).abs().squeeze(-1)[valid].max().item()
    dh = (hidden.cpu() - rh).abs()[valid].max().item()
    print("f16x3 ragged padded: logits %.2e hidden %.2e" % (dl, dh))
    assert dl < TOL and dh < TOL


def test_overlapped_host_streaming_matches_direct_scoring(vsa):
    """corpus.score_host_batches (H2D of batch i+1 under the kernels of batch i, two alternating device buffers):
    same scores as scoring each batch directly, including ragged batches with masks."""
    import importlib
    corpus = importlib.import_module("video-summarization_amd.corpus")
    synth = vsa.synth
    sd = synth.make_state_dict(256, 2, 31)
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=2, sparsity=0.0, dropout=0.3)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval()
    batches = []
    for i in range(5):
        lengths = [200 - 7 * i, 150, 33 + i]
        x = synth.make_features(3, 200 - 7 * i, 40 + i, "pool5", lengths=lengths)
        batches.append((x.pin_memory() if i % 2 == 0 else x, synth.padding_mask(x)))
    with torch.no_grad():
        outs = corpus.score_host_batches(lambda xx, mm: m.score(xx, mm), batches, _dev())
        for (x, mask), o in zip(batches, outs):
            want = m.score(x.to(_dev()), mask.to(_dev())).cpu()
            valid = ~mask
            assert torch.equal(o[valid], want[valid])
        # packed batches through the same double-buffered stream
        packed = []
        for i in range(4):
            lengths = [150 + 3 * i, 64, 201]
            packed.append((torch.cat([synth.make_features(1, t, 70 + 10 * i + j, "pool5")[0] for j, t in enumerate(lengths)]), lengths))
        outs = corpus.score_host_batches(lambda xx, ln: m.score_packed(xx, ln), packed, _dev())
        for (xp, lengths), o in zip(packed, outs):
            assert torch.equal(o, m.score_packed(xp.to(_dev()), lengths).cpu())


def test_randomised_parity_slice(vsa):
    """A fixed-seed slice of tests/fuzz_parity.py (random architecture, batch, lengths, mask kind; every compute
    mode; tiled and default dispatch) against the oracle.  The full soak (thousands of cases) is run by hand."""
    import importlib.util
    import os as _os
    spec = importlib.util.spec_from_file_location(
        "fuzz_parity", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, worst = mod.run(budget=60.0, seed=7, max_cases=40)
    print("fuzz slice: %d cases, worst %s" % (n, worst))
    assert n == 40


@pytest.mark.parametrize("compute", ["fp32", "fp16x3"])
@pytest.mark.parametrize("cfg", [(4, 256, 4), (8, 256, 2), (4, 128, 2), (4, 512, 2)])      # (4, 512): head dim 128, M-B (round 4)
def test_packed_ragged_batch_is_bit_identical_to_scoring_each_video_alone(vsa, cfg, compute):
    """SimNet.forward_packed (frames of all videos concatenated, no padding rows, no mask): every video's logits
    and hidden state equal scoring that video alone bit for bit, and meet the oracle at 1e-4."""
    H, d, L = cfg
    synth = vsa.synth
    sd = synth.make_state_dict(d, L, 17)
    m = vsa.SimNet(num_heads=H, d_model=d, num_layers=L, sparsity=0.0, dropout=0.3)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval().set_compute_dtype(compute)
    lengths = [320, 1, 257, 64, 650, 33, 128, 129, 2000]
    vids = [synth.make_features(1, t, 100 + i, "pool5")[0] for i, t in enumerate(lengths)]
    x = torch.cat(vids, dim=0).to(_dev())
    with torch.no_grad():
        logits, hidden = m.forward_packed(x, lengths)
        sc = m.score_packed(x, lengths)
        row = 0
        for i, (v, t) in enumerate(zip(vids, lengths)):
            l1, h1 = m(v[None].to(_dev()))
            assert torch.equal(logits[row:row + t], l1[0]) and torch.equal(hidden[row:row + t], h1[0]), i
            assert (sc[row:row + t] - torch.sigmoid(l1[0, :, 0])).abs().max().item() < 1e-6
            if i in (0, 4):
                rl, rh = oracle_forward(sd, v[None], None, H)
                assert (l1.cpu() - rl).abs().max().item() < TOL and (h1.cpu() - rh).abs().max().item() < TOL
            row += t
    with pytest.raises(RuntimeError):
        m.forward_packed(x, lengths[:-1])                     # row count does not match
    with pytest.raises(RuntimeError):
        m.forward_packed(torch.cat([x, x[:1]]), lengths[:-1] + [2001])     # beyond the positional table


def test_packed_batches_head_dim_128_bf16_mode(vsa, lp_linear_everywhere):
    """M-B (head dim 128) packed in the bf16 compute mode: the 8-wave bf16 attention over 256-row work items, every video
    within the bf16 mode's tolerance of the oracle and of the same video scored alone."""
    synth = vsa.synth
    sd = synth.make_state_dict(512, 2, 19)
    m = vsa.SimNet(num_heads=4, d_model=512, num_layers=2, sparsity=0.0, dropout=0.3)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval().set_compute_dtype("bf16")
    assert m.attention_dtype == "bf16"
    lengths = [320, 1, 257, 700, 33]
    vids = [synth.make_features(1, t, 300 + i, "pool5")[0] for i, t in enumerate(lengths)]
    x = torch.cat(vids, dim=0).to(_dev())
    with torch.no_grad():
        logits, _hidden = m.forward_packed(x, lengths)
        row = 0
        for i, (v, t) in enumerate(zip(vids, lengths)):
            l1, _h1 = m(v[None].to(_dev()))
            assert (logits[row:row + t] - l1[0]).abs().max().item() < 2e-2, i
            if i in (0, 3):
                rl, _rh = oracle_forward(sd, v[None], None, 4)
                e = (logits[row:row + t].cpu() - rl[0]).abs().max().item()
                assert 1e-5 < e < 3e-2, (i, e)
            row += t


@pytest.mark.parametrize("compute", ["fp32", "fp16x3", "bf16"])
@pytest.mark.parametrize("cfg", [(8, 256, 1), (4, 128, 1), (4, 256, 1)])
def test_packed_workspace_is_never_overrun(vsa, cfg, compute):
    """vs_scorer_workspace_bytes_packed sizes the plan area for the 128-row tiling whatever tiling the flags of the
    forward select (head dim 32 with a low-precision attention plans 128-row tiles where the size query used to assume
    256): a guard region right behind a workspace of exactly `need` bytes must stay untouched."""
    H, d, L = cfg
    lib = vsa._lib.load()
    m = vsa.SimNet(num_heads=H, d_model=d, num_layers=L, sparsity=0.0, dropout=0.3)
    m.load_state_dict(vsa.synth.make_state_dict(d, L, 9), strict=True)
    m = m.to(_dev()).eval().set_compute_dtype(compute)
    lengths = [1024] * 16
    M = sum(lengths)
    x = torch.randn(M, 1024, device=_dev())
    handle = m._packed_weights(x.device).handle
    host = (C.c_int32 * len(lengths))(*lengths)
    dev_len = torch.tensor(lengths, dtype=torch.int32, device=_dev())
    need = lib.vs_scorer_workspace_bytes_packed(handle, host, len(lengths))
    assert need > 0 and need % 256 == 0
    guard = 1 << 16
    buf = torch.full((need + guard,), 0xA5, dtype=torch.uint8, device=_dev())
    scores = torch.empty(M, 1, device=_dev())
    flags = m._attention_flag()
    vsa._lib.check(lib.vs_scorer_forward_packed(handle, x.data_ptr(), host, dev_len.data_ptr(), len(lengths), flags,
                                                scores.data_ptr(), None, buf.data_ptr(), need, _stream()))
    torch.cuda.synchronize()
    assert bool((buf[need:] == 0xA5).all()), "the forward wrote past its workspace"
    assert torch.isfinite(scores).all()
    # a workspace one byte short is refused
    rc = lib.vs_scorer_forward_packed(handle, x.data_ptr(), host, dev_len.data_ptr(), len(lengths), flags,
                                      scores.data_ptr(), None, buf.data_ptr(), need - 1, _stream())
    assert rc == vsa._lib.VS_ERR_WORKSPACE


def test_class_token_model_matches_reference_golden_and_oracle(vsa):
    """use_cls=True (reference simnet.py:47-51, 205-206, 214-216; no reference caller enables it): T + 1 output rows.
    Mask-free calls against vectors from the imported reference; masked calls against the oracle (the reference's
    own mask branch needs a CUDA device, simnet.py:49, so it cannot produce vectors in the build container)."""
    import json
    import os
    import numpy as np
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "cls_golden.npz"))
    for c in json.loads(str(z["cases"])):
        sd = vsa.synth.make_state_dict(c["d"], c["L"], c["wseed"], use_cls=True)
        m = vsa.SimNet(num_heads=c["H"], d_model=c["d"], num_layers=c["L"], sparsity=0.0, use_cls=True, dropout=0.3)
        m.load_state_dict(sd, strict=True)
        m = m.to(_dev()).eval()
        x = vsa.synth.make_features(c["B"], c["T"], c["xseed"], "randn")
        with torch.no_grad():
            logits, hidden = m(x.to(_dev()))
            assert logits.shape == (c["B"], c["T"] + 1, 1) and hidden.shape == (c["B"], c["T"] + 1, c["d"])
            assert (logits.cpu() - torch.from_numpy(z[c["name"] + ":logits"])).abs().max().item() < TOL
            assert (hidden.cpu() - torch.from_numpy(z[c["name"] + ":hidden"])).abs().max().item() < TOL
            # with a key mask (the class token itself is never masked)
            mask = vsa.synth.random_mask(c["B"], c["T"], 3)
            lm, hm = m(x.to(_dev()), mask.to(_dev()))
            rl, rh = oracle_forward(sd, x, mask, c["H"])
            assert (lm.cpu() - rl).abs().max().item() < TOL and (hm.cpu() - rh).abs().max().item() < TOL
            s = m.score(x.to(_dev()), mask.to(_dev()))
            assert s.shape == (c["B"], c["T"] + 1) and (s.cpu() - torch.sigmoid(rl.squeeze(-1))).abs().max().item() < TOL
        with pytest.raises(NotImplementedError):
            m.train()(x.to(_dev()))                          # training with a class token is not offered
        with pytest.raises(NotImplementedError):
            m.score_packed(x[0].to(_dev()), [c["T"]])


def test_class_token_scoring_launches_no_torch_kernels_after_the_first_call(vsa):
    """``use_cls`` scoring goes through ONE C call (``vs_scorer_forward_cls``) on the cached packed
    weights - after the first call (which packs them) a forward launches no ``at::native`` kernel (no per-call weight
    cast / ``torch.cat``) and no device-to-device memcpy: only kernels of libvsscore.so.  Also: the low-precision modes
    run with a class token (they did not before the token moved into the library)."""
    from torch.profiler import ProfilerActivity, profile
    sd = vsa.synth.make_state_dict(256, 2, 21, use_cls=True)
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=2, sparsity=0.0, use_cls=True, dropout=0.0)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval()
    x = vsa.synth.make_features(2, 150, 7, "randn").to(_dev())
    mask = vsa.synth.random_mask(2, 150, 5).to(_dev())
    with torch.no_grad():
        ref_l, ref_h = m(x, mask)                             # first call: packs the weights
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            l2, h2 = m(x, mask)
            s2 = m.score(x, mask)
            torch.cuda.synchronize()
    names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    assert names, "no device activity recorded"
    bad = [n for n in names if "at::native" in n or "Memcpy" in n or "Memset" in n]
    assert not bad, bad
    assert torch.equal(l2, ref_l) and torch.equal(h2, ref_h)
    assert (s2 - torch.sigmoid(ref_l.squeeze(-1))).abs().max().item() < 1e-6
    rl, rh = oracle_forward(sd, x.cpu(), mask.cpu(), 4)
    assert (ref_l.cpu() - rl).abs().max().item() < TOL and (ref_h.cpu() - rh).abs().max().item() < TOL
    with torch.no_grad():
        m.set_compute_dtype("fp16x3")
        lf, _ = m(x, mask)
        assert (lf.cpu() - rl).abs().max().item() < TOL
        m.set_compute_dtype("bf16")
        lb, _ = m(x, mask)
        assert (lb.cpu() - rl).abs().max().item() < tol.BF16_LOGIT_TOL


def test_weight_images_built_on_one_stream_are_ordered_for_calls_on_another(vsa):
    """the kernel-layout images of the parameters are rebuilt lazily by the first call that needs them, on THAT
    call's stream, and the host-side version stamp is set at enqueue time.  A second call on another stream used to see
    the stamp as current and could read images whose pack kernels were still running.  The library now records an event
    behind every parameter write / rebuild and makes a call on another stream wait for it on the device."""
    sd = vsa.synth.make_state_dict(1024, 4, 9)
    m = vsa.SimNet(num_heads=16, d_model=1024, num_layers=4, sparsity=0.0, dropout=0.0)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval()
    x = vsa.synth.make_features(1, 96, 3, "randn").to(_dev())
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for mode in ("fp16x3", "bf16", "fp32"):
        m.set_compute_dtype(mode)
        with torch.no_grad():
            for rep in range(3):
                for p in m.parameters():
                    p.mul_(1.0 + 1e-4)                      # new parameter version: every image family is stale
                torch.cuda.synchronize()
                with torch.cuda.stream(s1):
                    y1 = m(x)[0]                            # parameter copy + image rebuild + forward on s1
                with torch.cuda.stream(s2):
                    y2 = m(x)[0]                            # same version: no rebuild - must wait for s1's images
                torch.cuda.synchronize()
                assert torch.isfinite(y2).all() and torch.equal(y1, y2), (mode, rep)


@pytest.mark.parametrize("name", ["mb_t320", "mb_pad_t150", "d768_h12_t200_pad", "d1024_h8_t150", "d1024_h16_randmask_t96"])
def test_bf16_mode_on_wide_models_matches_reference_golden(vsa, lp_linear_everywhere, name):
    """Round 3: ``set_compute_dtype("bf16")`` for d_model > 256 (M-B = the reference's argparse default, train.py:169-173;
    d_model 768 / 1024): plain bf16 GEMMs + the row LayerNorm pass, bf16 attention at head dim 64 / 128.  Against the
    vectors of the imported reference at the bf16 mode's stated tolerance."""
    case = [c for c in golden_cases() if c["name"] == name][0]
    sd, x, mask = build_case(vsa.synth, case)
    g = load_golden(name)
    m = vsa.SimNet(num_heads=case["H"], d_model=case["d"], num_layers=case["L"], sparsity=0.0, dropout=0.3)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval().set_compute_dtype("bf16")
    assert m.attention_dtype == "bf16" and m.linear_dtype == "bf16"
    with torch.no_grad():
        logits, hidden = m(x.to(_dev()), None if mask is None else mask.to(_dev()))
        exact = m.set_compute_dtype("fp32")(x.to(_dev()), None if mask is None else mask.to(_dev()))[0]
    valid = torch.ones(x.shape[:2], dtype=torch.bool) if mask is None else ~mask
    err = (logits.cpu() - g["logits"])[valid].abs().max().item()
    print("bf16 mode %s: max |logit - reference| %.2e" % (name, err))
    assert 1e-6 < err < tol.BF16_LOGIT_TOL
    assert (exact.cpu() - g["logits"])[valid].abs().max().item() < TOL
    # the two halves separately: bf16 Linears with the exact attention run gemm_nt_128's bf16 form (fp32 operands rounded on
    # their way into LDS) - a different kernel family from the bf16-operand GEMM of the full bf16 mode - and the bf16
    # attention with exact Linears reads fp32 q / k / v
    with torch.no_grad():
        for lin, att in (("bf16", "fp32"), ("fp32", "bf16")):
            m.linear_dtype, m.attention_dtype = lin, att
            mixed = m(x.to(_dev()), None if mask is None else mask.to(_dev()))[0]
            e2 = (mixed.cpu() - g["logits"])[valid].abs().max().item()
            assert 1e-7 < e2 < tol.BF16_LOGIT_TOL, (lin, att, e2)
            assert not torch.equal(mixed, logits)


# ---- round 4: shapes outside the kernels' own envelope, run EMBEDDED in the next supported shape (simnet.embedding_plan,
# vs_weights_set_norm_width).  The reference-generated goldens of these shapes ride in golden_cases() above; here: the other
# entry points of the module on such a model, against the CPU oracle.
@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(8, 128), (5, 200), (3, 96), (1, 200)], ids=lambda c: "h%d_d%d" % c)      # (1, 200): head dim 200 -> 256
def test_embedded_shape_through_every_entry_point(vsa, cfg):
    """SimNet(num_heads=8, d_model=128) (head dim 16), (5, 200) (head dim 40), (3, 96): padded and masked batches, packed ragged
    batches, score(), the class-token variant and the low-precision compute modes, each against the oracle on the TRUE-shaped
    state dict; `hidden` comes back with d_model columns."""
    H, d = cfg
    synth = vsa.synth
    sd = synth.make_state_dict(d, 2, 77 + d)
    m = vsa.SimNet(num_heads=H, d_model=d, num_layers=2, sparsity=0.0, dropout=0.3)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval()
    assert m._plan is not None and m._lib_d > d
    lengths = [150, 97, 33]
    x = synth.make_features(3, 150, 5, "pool5", lengths)
    mask = synth.padding_mask(x)
    rl, rh = oracle_forward(sd, x, mask, H)
    valid = ~mask
    with torch.no_grad():
        logits, hidden = m(x.to(_dev()), mask.to(_dev()))
        assert hidden.shape == (3, 150, d) and logits.shape == (3, 150, 1)
        assert (logits.cpu() - rl)[valid].abs().max().item() < TOL and (hidden.cpu() - rh)[valid].abs().max().item() < TOL
        sc = m.score(x.to(_dev()), mask.to(_dev()))
        assert (sc.cpu() - torch.sigmoid(rl.squeeze(-1)))[valid].abs().max().item() < TOL
        # packed ragged batch: no padding rows, each video as if scored alone (head dim 256 has no packed attention: it says so)
        xp = torch.cat([x[i, :t] for i, t in enumerate(lengths)]).to(_dev())
        if m._lib_dh == 256:
            with pytest.raises(RuntimeError, match="packed batches need head_dim"):
                m.forward_packed(xp, lengths)
        else:
            pl, ph = m.forward_packed(xp, lengths)
            assert ph.shape == (sum(lengths), d)
            o = 0
            for i, t in enumerate(lengths):
                al, ah = m(x[i:i + 1, :t].to(_dev()))
                assert torch.equal(pl[o:o + t], al[0]) and torch.equal(ph[o:o + t], ah[0]), "packed != alone (video %d)" % i
                assert (al.cpu() - rl[i:i + 1, :t]).abs().max().item() < TOL
                o += t
        # the emulated-fp32 mode stays inside the 1e-4 bar; the bf16 mode inside its own
        m.set_compute_dtype("fp16x3")
        l3, h3 = m(x.to(_dev()), mask.to(_dev()))
        assert (l3.cpu() - rl)[valid].abs().max().item() < TOL and (h3.cpu() - rh)[valid].abs().max().item() < TOL
        vsa._lib.set_option("VS_LP_MIN_ROWS", 0)
        try:
            m.set_compute_dtype("bf16")
            lb, hb = m(x.to(_dev()), mask.to(_dev()))
        finally:
            vsa._lib.set_option("VS_LP_MIN_ROWS", -1)
        eb = (lb.cpu() - rl)[valid].abs().max().item()
        assert 1e-5 < eb < 3e-2, eb
        m.set_compute_dtype("fp32")
    # class token (simnet.py:205-216): T + 1 positions
    sdc = synth.make_state_dict(d, 2, 78 + d, use_cls=True)
    mc = vsa.SimNet(num_heads=H, d_model=d, num_layers=2, sparsity=0.0, dropout=0.3, use_cls=True)
    mc.load_state_dict(sdc, strict=True)
    mc = mc.to(_dev()).eval()
    cl, ch = oracle_forward(sdc, x, mask, H)
    with torch.no_grad():
        gl, gh = mc(x.to(_dev()), mask.to(_dev()))
    v1 = torch.cat([torch.ones(3, 1, dtype=torch.bool), valid], dim=1)
    assert gh.shape == (3, 151, d)
    assert (gl.cpu() - cl)[v1].abs().max().item() < TOL and (gh.cpu() - ch)[v1].abs().max().item() < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ma_t320", "ma_t320_pad", "ma_ragged37", "ma_randmask_t96", "ma_nc3_t64", "ma_nopos_t129", "dh32_t65",
                                  "mb_t320", "mb_pad_t150", "ctor_default_t100", "d128_h1_t100"])      # + d_model 512 (head dim 128 / 64), 128
def test_latency_mode_meets_the_goldens_and_is_deterministic(vsa, name):
    """SimNet.set_latency_mode() (VS_FLAG_SPLITK: split-K embedding / out-projection / fc2 + row LayerNorm for reference-sized
    calls): the reference-generated goldens at the path's own 1e-4 bar, within 2e-5 of the default kernels, bitwise
    reproducible, and independent of the batch a video is scored in (fixed slices, fixed reduction order)."""
    case = [c for c in golden_cases() if c["name"] == name][0]
    g = load_golden(name)
    sd, x, mask = build_case(vsa.synth, case)
    m = _model(vsa, case, sd)
    xd, md = x.to(_dev()), None if mask is None else mask.to(_dev())
    with torch.no_grad():
        l0, h0 = m(xd, md)
        m.set_latency_mode(True)
        l1, h1 = m(xd, md)
        l2, h2 = m(xd, md)
        valid = torch.ones(x.shape[:2], dtype=torch.bool) if mask is None else ~mask
        dl = (l1.cpu() - g["logits"])[valid].abs().max().item()
        dh = (h1.cpu()[:, g["rows"]] - g["hidden"])[valid[:, g["rows"]]].abs().max().item()
        dd = (l1 - l0).cpu()[valid].abs().max().item()
        print("latency mode %s: logits %.2e hidden %.2e of the golden; %.2e of the default kernels" % (name, dl, dh, dd))
        assert dl < TOL and dh < TOL and dd < 2e-5
        assert torch.equal(l1, l2) and torch.equal(h1, h2)
        if x.shape[0] > 1 and mask is None:           # each video alone == the same video inside the batch
            a, _ = m(xd[:1])
            assert torch.equal(a[0], l1[0])
        if name in ("ma_t320", "mb_t320", "ctor_default_t100"):
            assert not torch.equal(l0, l1), "the latency mode did not run (results are the default kernels' bits)"
        # the same with the Linears emulated on the f16 pipe (set_compute_dtype("fp16x3")): the split-K kernels have that form too,
        # the attention stays the exact split-key kernel - still the fp32 path's 1e-4 bar
        m.set_compute_dtype("fp16x3")
        l3, h3 = m(xd, md)
        l4, _h4 = m(xd, md)
        m.set_latency_mode(False)
        l5, _h5 = m(xd, md)
        m.set_compute_dtype("fp32")
        d3 = (l3.cpu() - g["logits"])[valid].abs().max().item()
        dh3 = (h3.cpu()[:, g["rows"]] - g["hidden"])[valid[:, g["rows"]]].abs().max().item()
        assert d3 < TOL and dh3 < TOL and torch.equal(l3, l4), (d3, dh3)
        if name in ("ma_t320", "mb_t320"):
            assert not torch.equal(l3, l5), "fp16x3: the latency mode did not run"


@pytest.mark.gpu
def test_scoring_forward_is_stream_capturable(vsa):
    """The scoring forward only enqueues on the caller's stream (no synchronisation, no allocation of its own after the first
    call): it can be captured into a graph (torch.cuda.CUDAGraph = hipGraph) as it is, and the replay reproduces the eager
    result bit for bit - in the default kernels and in the latency mode."""
    synth = vsa.synth
    sd = synth.make_state_dict(256, 2, 5)
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=2, sparsity=0.0, dropout=0.3)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval()
    x = synth.make_features(1, 320, 3, "pool5").to(_dev())
    with torch.no_grad():
        for mode in (False, True):
            m.set_latency_mode(mode)
            ref_l, ref_h = (t.clone() for t in m(x))
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                m(x)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out_l, out_h = m(x)
            x.copy_(synth.make_features(1, 320, 4, "pool5").to(_dev()))       # new input in the captured buffer
            g.replay()
            torch.cuda.synchronize()
            want_l, want_h = m(x)
            assert torch.equal(out_l, want_l) and torch.equal(out_h, want_h), mode
            assert not torch.equal(out_l, ref_l)
            x.copy_(synth.make_features(1, 320, 3, "pool5").to(_dev()))
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out_l, ref_l) and torch.equal(out_h, ref_h), mode
            del g


@pytest.mark.gpu
@pytest.mark.parametrize("tool,seconds,seed", [("fuzz_attn_w64.py", 20, 5), ("fuzz_attention.py", 25, 6), ("fuzz_linear.py", 15, 7)])
def test_kernel_soak_slices(vsa, tool, seconds, seed):
    """Fixed-seed slices of the round-4 kernel soaks (tools/): the one-wave-per-SIMD attention with late dominant keys and key
    masks, every attention entry point x head dims, the Linear / Linear + LayerNorm entry points over random shapes - each
    against float64.  The soaks found two real bugs in round 4; the full runs are in profiles/."""
    import importlib.util
    import os as _os
    path = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tools", tool)
    spec = importlib.util.spec_from_file_location(tool[:-3], path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(float(seconds), seed)          # raises on the first violation
