"""Evaluation harness: the counterpart of the reference's ``val_step`` (``src/train.py:134-152``).

``val_step(model, loader, device)`` keeps the reference's contract: ``loader`` yields
``(feature [1,T,1024], target [1,T], user)`` per video (reference ``collate_fn_test``,
``data/dataset.py:164-168``), the model is called unchanged, ``sigmoid`` is applied by the caller and
the per-video scores go to ``eval_metrics`` keyed by ``user.name``.  Returns
``(mean MSE loss, f_score, kendall_tau, spearman_r)`` like the reference.

``val_step_batched`` is the MI355X-friendly form of the same computation: the videos are scored in
length-bucketed padded batches (optionally sharded over ranks, scores gathered with one all_gather) and the
result is bit-identical per frame, because a video's scores do not depend on the batch it is scored in.
"""
from __future__ import annotations

from typing import Iterable, Sequence

import torch
import torch.nn.functional as F

from .corpus import plan_shards, score_corpus
from .evaluation import eval_metrics, eval_videos


@torch.no_grad()
def val_step(model, loader: Iterable, device):
    model.eval()
    score_dict, user_dict = {}, {}
    loss_sum, n = 0.0, 0
    for feature, target, user in loader:
        feature, target = feature.to(device), target.to(device)
        pred, _ = model(feature)                                   # train.py:143
        pred = torch.sigmoid(pred.view(1, -1))                     # train.py:144
        loss_sum += F.mse_loss(pred, target).item()                # train.py:145-147
        n += 1
        score_dict[user.name] = pred.squeeze(0).detach().cpu().numpy()
        user_dict[user.name] = user
    f_score, ktau, spr = eval_metrics(score_dict, user_dict)       # train.py:150
    return loss_sum / max(n, 1), f_score, ktau, spr


def evaluate_shard(scores, targets, users, order):
    """[sum MSE loss, sum F-score, sum Kendall tau, sum Spearman rho] over the videos `order` (indices): the keyshot
    evaluation of a rank's shard as ONE library call (`evaluation.eval_videos` -> vs_eval_corpus: one bounded host thread
    pool over every video and every (video, user) pair), sums in index order."""
    if not order:
        return [0.0, 0.0, 0.0, 0.0]
    f, k, s = eval_videos({i: scores[i].numpy() for i in order}, {i: users[i] for i in order}, "avg")
    loss = 0.0
    for i in order:
        loss += F.mse_loss(scores[i].view(1, -1), targets[i].detach().float().cpu().view(1, -1)).item()
    return [loss, float(f.sum()), float(k.sum()), float(s.sum())]


@torch.no_grad()
def val_step_batched(model, features: Sequence[torch.Tensor], targets: Sequence[torch.Tensor], users: Sequence,
                     device, rank: int = 0, world: int = 1, group=None, max_frames: int = 65536):
    """Same result as ``val_step`` over (features[i] [T_i,1024], targets[i] [T_i], users[i])."""
    model.eval()
    # models with head dim 32 / 64 score PACKED batches (no sentinel padding, no mask; the same bits)
    can_pack = hasattr(model, "score_packed") and getattr(model, "_lib_dh", model.d_model // model.num_heads) in (32, 64, 128)
    scores = score_corpus(lambda x, m: model.score(x, m), list(features), rank=rank, world=world, group=group,
                          device=device, max_frames=max_frames,
                          packed_fn=(lambda x, lens: model.score_packed(x, lens)) if can_pack else None)
    # every rank holds every video's scores; the (CPU) evaluation is sharded too: a rank evaluates the videos it
    # scored and the four sums are all-reduced.  Sums run in a fixed per-rank order.
    lengths = [int(f.shape[0]) for f in features]
    mine = plan_shards(lengths, world)[rank] if world > 1 else list(range(len(users)))
    order = sorted(mine)

    sums = evaluate_shard(scores, targets, users, order)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor(sums, dtype=torch.float64,
                         device=device if (device is not None and dist.get_backend(group) == "nccl") else "cpu")
        dist.all_reduce(t, group=group)
        sums = t.tolist()
    n = max(len(users), 1)
    return sums[0] / n, sums[1] / n, sums[2] / n, sums[3] / n
