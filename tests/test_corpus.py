"""CPU: corpus sharding / batching host logic and the world_size-2 gather path (gloo)."""
import importlib
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _videos(n, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(12, 60, (n,), generator=g).tolist()
    return [torch.randn(t, 1024, generator=g) for t in lens]


def _oracle_score_fn(sd, H):
    from oracle.simnet_oracle import oracle_scores

    def fn(x, mask):
        with torch.no_grad():
            return oracle_scores(sd, x, mask, H)
    return fn


def test_plan_shards_covers_everything_once_and_balances(vsa):
    corpus = importlib.import_module("video-summarization_amd.corpus")
    lengths = [150 + (37 * i) % 500 for i in range(75)]          # 50 + 25 videos, cfg 4 shape
    for world in (1, 2, 3, 8):
        shards = corpus.plan_shards(lengths, world)
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(75))
        loads = [sum(corpus.video_cost(lengths[i]) for i in s) for s in shards]
        assert max(loads) / (sum(loads) / world) < 1.15
    assert corpus.plan_shards([], 4) == [[], [], [], []]


def test_bucket_batches_bounds_padding(vsa):
    corpus = importlib.import_module("video-summarization_amd.corpus")
    lengths = [100, 105, 110, 400, 410, 50, 1000]
    batches = corpus.bucket_batches(range(len(lengths)), lengths, max_frames=1200, max_waste=0.2)
    assert sorted(i for b in batches for i in b) == list(range(len(lengths)))
    for b in batches:
        tmax = max(lengths[i] for i in b)
        assert tmax * len(b) <= 1200 or len(b) == 1
        assert 1 - sum(lengths[i] for i in b) / (tmax * len(b)) <= 0.2 + 1e-9
    x, mask = corpus.pad_batch([torch.ones(3, 1024), torch.ones(5, 1024)])
    assert x.shape == (2, 5, 1024) and x[0, 3:, 0].eq(1000.0).all() and mask[0].tolist() == [False] * 3 + [True] * 2
    assert corpus.pad_batch([torch.ones(4, 1024), torch.ones(4, 1024)])[1] is None


def test_single_process_corpus_equals_per_video_scoring(vsa):
    corpus = importlib.import_module("video-summarization_amd.corpus")
    sd = vsa.synth.make_state_dict(256, 1, 5)
    vids = _videos(7, 3)
    fn = _oracle_score_fn(sd, 4)
    got = corpus.score_corpus(fn, vids, max_frames=150)
    for i, v in enumerate(vids):
        want = fn(v.unsqueeze(0), None)[0]
        assert (got[i] - want).abs().max().item() < 1e-5      # padded+masked batch vs alone


def _framewise_fn(x, mask):
    """Batch-invariant stand-in scorer (each frame's score depends on that frame only), so the
    sharding/gather plumbing can be held to bit-equality on CPU."""
    return x[..., 0] * 0.5 + x[..., 1]          # exactly-rounded elementwise ops only


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(2)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vsa = importlib.import_module("video-summarization_amd")
    corpus = importlib.import_module("video-summarization_amd.corpus")
    sd = vsa.synth.make_state_dict(256, 1, 5)
    res = corpus.score_corpus(_oracle_score_fn(sd, 4), _videos(9, 11), rank=rank, world=world, max_frames=150)
    res2 = corpus.score_corpus(_framewise_fn, _videos(9, 11), rank=rank, world=world, max_frames=150)
    # plain lists: torch tensors would travel as shared-memory handles that die with the worker
    q.put((rank, {k: v.tolist() for k, v in res.items()}, {k: v.tolist() for k, v in res2.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_rank_bit_for_bit(vsa):
    """N>1 path with world_size 2 over gloo: every rank ends up with every video's scores.  The
    plumbing (shard, pad, gather, un-pad) is exact — bit-for-bit with a batch-invariant scorer; with the
    CPU oracle as scorer the batch composition changes ATen's blocking, so that leg is held to 1e-5.
    (Bit-equality of the HIP scorer across shardings is a GPU test in test_hip_parity.py.)"""
    corpus = importlib.import_module("video-summarization_amd.corpus")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    results = {r: {k: torch.tensor(v) for k, v in a.items()} for r, a, _ in got}
    results2 = {r: {k: torch.tensor(v) for k, v in b.items()} for r, _, b in got}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sd = vsa.synth.make_state_dict(256, 1, 5)
    torch.set_num_threads(2)
    single = corpus.score_corpus(_oracle_score_fn(sd, 4), _videos(9, 11), max_frames=150)
    single2 = corpus.score_corpus(_framewise_fn, _videos(9, 11), max_frames=150)
    for r in (0, 1):
        assert sorted(results[r]) == list(range(9)) and sorted(results2[r]) == list(range(9))
        for i in range(9):
            assert torch.equal(results2[r][i], single2[i])
            assert torch.equal(results[r][i], results[1 - r][i])
            assert (results[r][i] - single[i]).abs().max().item() < 1e-5


class _StubScorer:
    """Frame-wise, batch-invariant stand-in for SimNet.score on CPU (the HIP scorer needs a GPU)."""
    def eval(self):
        return self

    def score(self, x, mask=None):
        return torch.sigmoid(x[..., 0] * 0.5 + x[..., 1])


def _val_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    harness = importlib.import_module("video-summarization_amd.harness")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    feats, targets, users = importlib.import_module("eval_corpus").corpus(seed=3)
    out = harness.val_step_batched(_StubScorer(), feats[:12], targets[:12], users[:12], None, rank, world, max_frames=2048)
    q.put((rank, [float(v) for v in out]))
    dist.barrier()
    dist.destroy_process_group()


def test_val_step_batched_two_ranks_equals_one_rank(vsa):
    """Sharded scoring + gather + keyshot evaluation (world_size 2, gloo) returns exactly the 1-rank metrics."""
    vsa._lib.build()
    harness = importlib.import_module("video-summarization_amd.harness")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    feats, targets, users = importlib.import_module("eval_corpus").corpus(seed=3)
    single = harness.val_step_batched(_StubScorer(), feats[:12], targets[:12], users[:12], None, max_frames=2048)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_val_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        assert got[r] == got[0]
        assert all(abs(a - b) < 1e-12 for a, b in zip(got[r], single))      # sums regrouped per rank: last-bit only
