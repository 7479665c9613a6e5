#!/usr/bin/env python3
"""Golden vectors for the keyshot evaluation, produced by IMPORTING the reference ``evaluation`` package
(reference src/evaluation/__init__.py:2) on CPU in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_eval.py

Synthetic TVSum-shaped records (fields of reference data/dataset.py:146-154; 5 videos named like split 0 of
splits_dsnet/tvsum.yaml: video_22, 7, 6, 11, 1).  Data only is stored: inputs and the reference's outputs."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("VS_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "src"))
sys.dont_write_bytecode = True


class Rec:                      # same attribute names as reference UserSummaries
    def __init__(self, **kw):
        self.__dict__.update(kw)


def make_video(rng, name, n_frames):
    picks = np.arange(0, n_frames, 15)
    # shots: random change points, inclusive ends, covering [0, n_frames)
    cuts = np.sort(rng.choice(np.arange(30, n_frames - 30), size=max(3, n_frames // 120), replace=False))
    starts = np.concatenate([[0], cuts])
    ends = np.concatenate([cuts - 1, [n_frames - 1]])
    cps = np.stack([starts, ends], axis=1).astype(np.int64)
    n_users = 20
    user_scores = rng.integers(1, 6, size=(n_users, n_frames)).astype(np.float32)      # TVSum 1..5 ratings
    user_scores = np.repeat(user_scores[:, ::40], 40, axis=1)[:, :n_frames]              # piecewise constant -> ties
    user_summary = (rng.random((n_users, n_frames)) < 0.15).astype(np.float32)
    scores = (1.0 / (1.0 + np.exp(-rng.standard_normal(len(picks))))).astype(np.float32)
    return scores, Rec(user_summary=user_summary, user_scores=user_scores, change_points=cps,
                       n_frames=n_frames, picks=picks, name=name)


def main():
    from evaluation import eval_metrics                                   # the reference
    from evaluation.compute_metrics import upsample
    from evaluation.generate_summary import generate_summary
    from evaluation.knapsack_implementation import knapSack
    from evaluation.evaluation_metrics import evaluate_summary
    from evaluation.compute_correlation import evaluate_scores
    rng = np.random.Generator(np.random.PCG64(2024))
    names = ["video_22", "video_7", "video_6", "video_11", "video_1"]
    frames = [4500, 3103, 9534, 2211, 6871]
    data, users, store = {}, {}, {}
    for i, (n, nf) in enumerate(zip(names, frames)):
        sc, u = make_video(rng, n, nf)
        data[n], users[n] = sc, u
        summ = generate_summary([u.change_points], [sc], [nf], [u.picks])[0]
        fs = upsample(sc, nf, u.picks)
        f = evaluate_summary(summ, u.user_summary, "avg")
        fmax = evaluate_summary(summ, u.user_summary, "max")
        k, s = evaluate_scores(fs, u.user_scores)
        store.update({"v%d_scores" % i: sc, "v%d_picks" % i: u.picks, "v%d_cps" % i: u.change_points,
                      "v%d_nframes" % i: nf, "v%d_user_summary" % i: u.user_summary.astype(np.int8),
                      "v%d_user_scores" % i: u.user_scores, "v%d_summary" % i: summ, "v%d_upsampled" % i: fs,
                      "v%d_metrics" % i: np.array([f, fmax, k, s])})
        print(n, "f %.4f fmax %.4f tau %.5f rho %.5f selected frames %d/%d" % (f, fmax, k, s, summ.sum(), len(summ)))
    store["eval_metrics"] = np.array(eval_metrics(data, users))
    # knapsack: the reference's own known-answer vector + random instances with float32-mean-like values
    assert knapSack(7, [2, 2, 1, 1, 1, 2], [4, 4, 2, 2, 2, 4], 6) == [0, 1, 2, 3, 4]
    for j in range(6):
        n = int(rng.integers(5, 60))
        wt = rng.integers(10, 300, size=n).tolist()
        val = [float(np.float32(x)) for x in rng.random(n)]
        W = int(sum(wt) * 0.15)
        store.update({"k%d_wt" % j: np.array(wt), "k%d_val" % j: np.array(val), "k%d_W" % j: W,
                      "k%d_sel" % j: np.array(knapSack(W, wt, val, n), dtype=np.int64)})
    # float32 shot means over many run lengths (pins the numpy pairwise summation order)
    x = rng.random(5000).astype(np.float32)
    lens = [1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 127, 128, 129, 130, 255, 256, 257, 300, 511, 777, 1024, 1500, 4999]
    store["mean_x"] = x
    store["mean_lens"] = np.array(lens)
    store["mean_vals"] = np.array([x[3:3 + L].mean().item() for L in lens if 3 + L <= 5000] )
    np.savez_compressed(os.path.join(HERE, "eval_golden.npz"), **store)
    print("eval_metrics", store["eval_metrics"])


if __name__ == "__main__":
    main()
