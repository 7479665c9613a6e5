"""Keyshot evaluation of the scorer's output — drop-in for the reference ``evaluation`` package.

Same call surface as reference ``src/evaluation/compute_metrics.py:42`` ``eval_metrics(data, user_dict)``
(called from ``train.py:150``): ``data`` maps video name -> per-(sub-sampled)-frame scores, ``user_dict``
maps video name -> a record with ``user_summary, user_scores, change_points, n_frames, picks``
(reference ``data/dataset.py:146-154``).  The work (up-sampling, float32 shot means, 0/1 knapsack,
F-score, Kendall tau / Spearman rho) runs in host C++ (``csrc/vs_eval.cpp``) through the C ABI of
``include/vs_eval.h``; there is no Python fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _i32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def upsample(scores, n_frames, positions) -> np.ndarray:
    """compute_metrics.py:19-39."""
    lib = _lib.load()
    s, pos = _f32(scores), _i32(positions)
    out = np.empty(int(n_frames), dtype=np.float32)
    _lib.check(lib.vs_eval_upsample(_p(s), s.size, _p(pos), pos.size, int(n_frames), _p(out)))
    return out


def knapSack(W, wt, val, n):
    """knapsack_implementation.py:1-30 -> list of selected shot indices."""
    lib = _lib.load()
    w = _i32(wt)[:n]
    v = np.ascontiguousarray(np.asarray(val, dtype=np.float64)[:n])
    sel = np.empty(max(n, 1), dtype=np.int32)
    cnt = C.c_int32()
    _lib.check(lib.vs_eval_knapsack(int(W), _p(w), _p(v), int(n), _p(sel), C.byref(cnt)))
    return sel[: cnt.value].tolist()


def generate_summary(all_shot_bound, all_scores, all_nframes, all_positions):
    """generate_summary.py:6-57 -> list of int8 summaries, one per video."""
    lib = _lib.load()
    out = []
    for sb, sc, nf, pos in zip(all_shot_bound, all_scores, all_nframes, all_positions):
        sb, sc, pos = _i32(sb), _f32(sc), _i32(pos)
        n = int(sb[-1, 1]) + 1
        summary = np.empty(n, dtype=np.int8)
        _lib.check(lib.vs_eval_generate_summary(_p(sc), sc.size, _p(pos), pos.size, int(nf), _p(sb), sb.shape[0],
                                                _p(summary), n))
        out.append(summary)
    return out


def evaluate_summary(predicted_summary, user_summary, eval_method):
    """evaluation_metrics.py:4-33."""
    lib = _lib.load()
    s = np.ascontiguousarray(np.asarray(predicted_summary), dtype=np.int8)
    us = np.ascontiguousarray(np.asarray(user_summary), dtype=np.int8)
    f = C.c_double()
    _lib.check(lib.vs_eval_fscore(_p(s), s.size, _p(us), us.shape[0], us.shape[1], 1 if eval_method == "max" else 0,
                                  C.byref(f)))
    return f.value


def evaluate_scores(predicted_scores, user_scores):
    """compute_correlation.py:4-15 -> (mean Kendall tau, mean Spearman rho)."""
    lib = _lib.load()
    ps = _f32(predicted_scores)
    us = np.ascontiguousarray(np.asarray(user_scores), dtype=np.float64)
    if us.shape[1] != ps.size:
        raise ValueError("user_scores has %d frames, prediction %d" % (us.shape[1], ps.size))
    k, s = C.c_double(), C.c_double()
    _lib.check(lib.vs_eval_rank_correlation(_p(ps), ps.size, _p(us), us.shape[0], C.byref(k), C.byref(s)))
    return k.value, s.value


def eval_videos(data, user_dict, eval_method="avg", max_threads=0):
    """The per-video results of eval_metrics: (f_score, kendall, spearman) arrays in the key order of `data`.
    ONE C call (vs_eval_corpus): every video and every (video, user) rank correlation runs on one bounded pool of host
    threads inside the library - no Python per video in the timed part, no nested pools."""
    lib = _lib.load()
    keys = list(data.keys())
    n = len(keys)
    recs = (_lib.EvalVideo * max(n, 1))()
    keep = []                                       # the arrays the records point into
    for j, k in enumerate(keys):
        u = user_dict[k]
        sc, pos, cp = _f32(data[k]).reshape(-1), _i32(u.picks).reshape(-1), _i32(u.change_points)
        us = np.ascontiguousarray(np.asarray(u.user_summary), dtype=np.int8)
        uf = np.asarray(u.user_scores)
        uf = np.ascontiguousarray(uf, dtype=np.float32 if uf.dtype == np.float32 else np.float64)      # float32 (the datasets' type) goes in as it is
        if uf.ndim != 2 or uf.shape[1] != int(u.n_frames):
            raise ValueError("user_scores of %r has shape %r, n_frames %d" % (k, uf.shape, int(u.n_frames)))
        keep.append((sc, pos, cp, us, uf))
        r = recs[j]
        r.scores, r.positions, r.change_points, r.user_summary, r.user_scores = _p(sc), _p(pos), _p(cp), _p(us), _p(uf)
        r.n_scores, r.n_positions, r.n_frames, r.n_shots = sc.size, pos.size, int(u.n_frames), cp.shape[0]
        r.n_users, r.user_len, r.n_score_users, r.use_max = us.shape[0], us.shape[1], uf.shape[0], 1 if eval_method == "max" else 0
        r.user_scores_f32 = 1 if uf.dtype == np.float32 else 0
    f, kt, sp = (np.empty(n, dtype=np.float64) for _ in range(3))
    _lib.check(lib.vs_eval_corpus(recs, n, int(max_threads), _p(f), _p(kt), _p(sp)))
    return f, kt, sp


def eval_metrics(data, user_dict):
    """compute_metrics.py:42-92 -> (mean F-score ['avg' protocol, :43], mean Kendall tau, mean Spearman rho);
    the means are taken in key order (the result does not depend on the threads' scheduling)."""
    if not len(data):
        return float(np.mean(())), float(np.mean(())), float(np.mean(()))
    f, kt, sp = eval_videos(data, user_dict, "avg")
    return float(np.mean(f)), float(np.mean(kt)), float(np.mean(sp))
