"""GPU tests at the FULL sizes of BASELINE.json's configs (the goldens cover them at reduced sizes):
configs[3] — the 75-video TVSum+SumMe-shaped corpus, packed and padded, sharded 8 ways; configs[4] — B=8, T=8192,
D=2048 (all three compute modes) through size-independent properties; and the host-side suites (keyshot evaluation)
run once more on the GPU box, where the round-end driver only collects `-m gpu` tests."""
import importlib

import numpy as np
import pytest
import torch

from oracle.simnet_oracle import oracle_forward
import tolerances as tol

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _dev():
    assert torch.cuda.is_available(), "these tests need a HIP device"
    return torch.device("cuda:0")


def _corpus_lengths():
    """SURVEY §8(d) cfg 4: 50 videos T~U[150,650] (TVSum) + 25 videos T~U[100,650] (SumMe), seeded."""
    rng = np.random.Generator(np.random.PCG64(75))
    return [int(t) for t in rng.integers(150, 651, 50)] + [int(t) for t in rng.integers(100, 651, 25)]


def test_config3_full_corpus_packed_padded_and_sharded(vsa):
    corpus = importlib.import_module("video-summarization_amd.corpus")
    lengths = _corpus_lengths()
    assert len(lengths) == 75
    rng = np.random.Generator(np.random.PCG64(76))
    videos = [torch.from_numpy((np.abs(rng.standard_normal((t, 1024))) * 0.5).astype(np.float32)) for t in lengths]
    sd = vsa.synth.make_state_dict(256, 4, 11)
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=4, sparsity=0.0, dropout=0.3)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval()
    padded = corpus.score_corpus(lambda x, mk: m.score(x, mk), videos, device=_dev(), max_frames=16384)
    packed = corpus.score_corpus(lambda x, mk: m.score(x, mk), videos, device=_dev(), max_frames=16384,
                                 packed_fn=lambda x, lens: m.score_packed(x, lens))
    assert sorted(padded) == sorted(packed) == list(range(75))
    for i in range(75):
        assert padded[i].shape == (lengths[i],) and torch.isfinite(padded[i]).all()
        assert torch.equal(padded[i], packed[i]), i                  # a video's scores do not depend on how it is batched
    # the 8-way shard plan of configs[3]: disjoint, complete, and every rank's shard scored on its own gives the same bits
    shards = corpus.plan_shards(lengths, 8)
    assert sorted(i for s in shards for i in s) == list(range(75))
    cost = [sum(corpus.video_cost(lengths[i]) for i in s) for s in shards]
    assert max(cost) / (sum(cost) / 8) < 1.08                        # balanced to a few percent
    for r in (0, 3, 7):
        idx = shards[r]
        part = corpus.score_corpus(lambda x, mk: m.score(x, mk), [videos[i] for i in idx], device=_dev(), max_frames=16384,
                                   packed_fn=lambda x, lens: m.score_packed(x, lens))
        for j, i in enumerate(idx):
            assert torch.equal(part[j], packed[i])
    # the oracle on a sample (shortest, median, longest)
    order = sorted(range(75), key=lambda i: lengths[i])
    for i in (order[0], order[37], order[-1]):
        rl, _ = oracle_forward(sd, videos[i][None], None, 4)
        assert (packed[i] - torch.sigmoid(rl[0, :, 0])).abs().max().item() < TOL


@pytest.mark.parametrize("compute", ["fp32", "fp16x3", "bf16"])
def test_config4_full_size_properties(vsa, compute):
    """B=8, T=8192, D=2048 (CLIP-ViT-shaped features; outside the reference's envelope, so the module is
    re-parameterised: in_features=2048, an 8192-row positional table).  No CPU oracle finishes at this size; the
    properties: finite; a video scored alone equals its row of the batch (bit for bit in fp32 / fp16x3); a padded
    copy of a shorter video scores its valid frames like the unpadded video; the low-precision modes stay within their
    stated distance of the exact path (tests/tolerances.py: bf16 BF16_LOGIT_TOL on logits, fp16x3 FP32_TOL)."""
    B, T, D = 8, 8192, 2048
    sd = vsa.synth.make_state_dict(256, 4, 5, in_features=D, max_len=T)
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=4, sparsity=0.0, dropout=0.3, in_features=D, pe_len=T)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval()
    g = torch.Generator(device="cpu").manual_seed(8192)
    x = torch.randn(B, T, D, generator=g).to(_dev())
    with torch.no_grad():
        exact, _ = m(x)
        m.set_compute_dtype(compute)
        full, hid = m(x)
        assert torch.isfinite(full).all() and torch.isfinite(hid).all()
        alone, _ = m(x[3:4].contiguous())
        if compute == "bf16":
            assert (alone - full[3:4]).abs().max().item() < tol.BF16_LOGIT_TOL       # bf16 Linears change kernel family with the row count
        else:
            assert torch.equal(alone, full[3:4])
        short = x[:1, :5000].contiguous()
        padded = torch.full((1, T, D), 1000.0, device=_dev())
        padded[:, :5000] = short
        a, _ = m(padded, padded[:, :, 0] == 1000.0)
        b, _ = m(short)
        assert (a[:, :5000] - b).abs().max().item() < (tol.BF16_LOGIT_TOL if compute == "bf16" else 5e-5)
        dist = (full - exact).abs().max().item()
        assert dist < {"fp32": 1e-12, "fp16x3": tol.FP32_TOL, "bf16": tol.BF16_LOGIT_TOL}[compute], dist


def test_host_side_suites_on_the_gpu_box(vsa):
    """tests/test_evaluation.py is CPU-marked (host C++ through the C ABI); run its cases here once more so the
    `-m gpu` collection of the round-end driver exercises vs_eval.cpp on the GPU box's CPU too."""
    te = importlib.import_module("test_evaluation")
    ev = importlib.import_module("video-summarization_amd.evaluation")
    te.test_knapsack_reference_known_answer(ev)
    for j in range(6):
        te.test_knapsack_matches_reference(ev, j)
    te.test_float32_shot_means_follow_numpy_pairwise_order(ev)
    for i in range(5):
        te.test_per_video_pipeline_matches_reference(ev, i)
    te.test_eval_metrics_drop_in(ev)
    te.test_edge_cases(ev)
    for j in range(8):
        te.test_knapsack_nan_values_follow_python_max(ev, j)
    for j in range(4):
        te.test_shots_past_n_frames_have_nan_means_like_the_reference(ev, j)


def test_three_times_the_bench_batch_scoring_and_training(vsa):
    """B=192, T=1024 (3x the bench batch: 196 608 frames, 0.8 GB of features, ~10 GB of training activations): index
    arithmetic beyond 2^31 bytes in every kernel family.  Scoring: finite, and rows 100..103 equal the same videos
    scored alone (bit for bit).  Training (dropout 0.3): finite loss and gradients, and the gradient of a batch made
    of ONE video repeated is, for every parameter, 192x the single-video gradient only in expectation - so instead the
    check is bitwise reproducibility of the whole step at this size."""
    B, T = 192, 1024
    sd = vsa.synth.make_state_dict(256, 4, 7)
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=4, sparsity=0.0, dropout=0.3)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev())
    g = torch.Generator(device="cpu").manual_seed(192)
    x = torch.randn(B, T, 1024, generator=g).to(_dev())
    with torch.no_grad():
        full, hid = m.eval()(x)
        part, _ = m(x[100:104].contiguous())
    assert torch.isfinite(full).all() and torch.isfinite(hid).all() and torch.equal(full[100:104], part)
    target = torch.rand(B, T, generator=g).to(_dev())
    mask = torch.zeros(B, T, dtype=torch.bool, device=_dev())
    mask[5, 900:] = True
    m.train()
    grads = []
    for _ in range(2):
        torch.manual_seed(99)
        m.zero_grad(set_to_none=True)
        pred, _ = m(x, mask)
        loss = vsa.mse_with_mask_loss(pred, target, mask)
        loss.backward()
        assert torch.isfinite(loss)
        grads.append([p.grad.clone() for p in m.parameters()])
    for a, b in zip(*grads):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_bf16_layer_kernels_with_more_row_tiles_than_cus(vsa):
    """B=80, T=1024 (81 920 rows = 320 row tiles of 256: more than the 256 CUs, so some blocks of the persistent bf16
    layer kernels - embedding + QKV, layer tail + next QKV - walk a second tile and restart their weight ring): the whole
    forward must equal, bit for bit, the same mode with those kernels switched back to the stand-alone ones."""
    B, T = 80, 1024
    sd = vsa.synth.make_state_dict(256, 2, 17, trained_like=True)
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=2, sparsity=0.0, dropout=0.3)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval().set_compute_dtype("bf16")
    g = torch.Generator(device="cpu").manual_seed(80)
    x = (torch.randn(B, T, 1024, generator=g).abs() * 0.5).to(_dev())
    mask = torch.zeros(B, T, dtype=torch.bool, device=_dev())
    mask[7, 1000:] = True
    mask[79, 500:] = True

    def run():
        with torch.no_grad():
            return [t.clone() for t in m(x, mask)]

    names = ("VS_LP_TAIL_UNFUSED", "VS_LP_QKV_UNFUSED", "VS_LP_EMBED_UNFUSED")
    try:
        for n in names:
            vsa._lib.set_option(n, 1)
        ref = run()
    finally:
        for n in names:
            vsa._lib.set_option(n, -1)
    got = run()
    for a, b in zip(got, ref):
        assert torch.isfinite(a).all() and torch.equal(a, b)
