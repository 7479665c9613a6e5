"""ORACLE — test infrastructure, not product code.

numpy/scipy restatement of the reference keyshot evaluation (reference ``src/evaluation/``), used only by
``tests/`` to check ``csrc/vs_eval.cpp``.  Pinned by the reference itself: ``tests/golden/make_golden_eval.py``
imports the reference ``evaluation`` package in the build container and commits its outputs
(``tests/golden/eval_*.npz``); the knapsack also has the reference's own known-answer vector
(``knapsack_implementation.py:36-41`` -> ``[0, 1, 2, 3, 4]``).
"""
import numpy as np
from scipy import stats


def upsample(scores, n_frames, positions):                          # compute_metrics.py:19-39
    frame_scores = np.zeros(n_frames, dtype=np.float32)
    positions = np.asarray(positions).astype(np.int32)
    if positions[-1] != n_frames:
        positions = np.concatenate([positions, [n_frames]])
    for i in range(len(positions) - 1):
        frame_scores[positions[i]:positions[i + 1]] = 0 if i == len(scores) else scores[i]
    return frame_scores


def knapsack(W, wt, val, n):                                        # knapsack_implementation.py:1-30
    K = [[0] * (W + 1) for _ in range(n + 1)]
    for i in range(1, n + 1):
        for w in range(1, W + 1):
            K[i][w] = max(val[i - 1] + K[i - 1][w - wt[i - 1]], K[i - 1][w]) if wt[i - 1] <= w else K[i - 1][w]
    sel, w = [], W
    for i in range(n, 0, -1):
        if K[i][w] != K[i - 1][w]:
            sel.insert(0, i - 1)
            w -= wt[i - 1]
    return sel


def generate_summary(shot_bound, scores, n_frames, positions):      # generate_summary.py:17-55 (one video)
    fs = upsample(scores, n_frames, positions)
    lengths = [int(s[1] - s[0] + 1) for s in shot_bound]
    imp = [fs[s[0]:s[1] + 1].mean().item() for s in shot_bound]
    last = shot_bound[-1]
    sel = knapsack(int((last[1] + 1) * 0.15), lengths, imp, len(lengths))
    summary = np.zeros(last[1] + 1, dtype=np.int8)
    for s in sel:
        summary[shot_bound[s][0]:shot_bound[s][1] + 1] = 1
    return summary


def fscore(pred, user_summary, method="avg"):                       # evaluation_metrics.py:4-33
    L = max(len(pred), user_summary.shape[1])
    S = np.zeros(L, dtype=int)
    S[:len(pred)] = pred
    out = []
    for u in range(user_summary.shape[0]):
        G = np.zeros(L, dtype=int)
        G[:user_summary.shape[1]] = user_summary[u]
        ov = (S & G).sum()
        p, r = ov / S.sum(), ov / G.sum()
        out.append(0 if p + r == 0 else 2 * p * r * 100 / (p + r))
    return max(out) if method == "max" else sum(out) / len(out)


def rank_correlation(frame_scores, user_scores):                    # compute_correlation.py:4-15
    k, s = [], []
    for u in user_scores:
        a, b = stats.rankdata(-frame_scores), stats.rankdata(-u)
        s.append(stats.spearmanr(a, b)[0])
        k.append(stats.kendalltau(a, b)[0])
    return sum(k) / len(k), sum(s) / len(s)
