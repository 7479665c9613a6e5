#!/usr/bin/env python3
"""Golden vectors for the NaN corner of the keyshot selection, produced by IMPORTING the reference ``evaluation``
package on CPU in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_eval_nan.py

A shot whose change points lie past ``n_frames`` averages an empty slice (reference generate_summary.py:42: NaN,
numpy RuntimeWarning) and the knapsack then takes ``max(NaN, x)`` / ``max(x, NaN)`` with Python semantics
(knapsack_implementation.py:18).  Data only is stored: inputs and the reference's outputs."""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("VS_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "src"))
sys.dont_write_bytecode = True


def main():
    from evaluation.generate_summary import generate_summary        # the reference
    from evaluation.knapsack_implementation import knapSack
    warnings.simplefilter("ignore")
    rng = np.random.Generator(np.random.PCG64(77))
    store = {}
    # knapsack instances with NaN values at random places
    for j in range(8):
        n = int(rng.integers(4, 30))
        wt = rng.integers(5, 120, size=n).tolist()
        val = [float(np.float32(x)) for x in rng.random(n)]
        for i in rng.choice(n, size=int(rng.integers(1, 4)), replace=False):
            val[int(i)] = float("nan")
        W = int(sum(wt) * (0.15 + 0.1 * j))
        store.update({"k%d_wt" % j: np.array(wt), "k%d_val" % j: np.array(val), "k%d_W" % j: W,
                      "k%d_sel" % j: np.array(knapSack(W, wt, val, n), dtype=np.int64)})
    # videos whose LAST shots start past n_frames (their mean is NaN)
    for j in range(4):
        n_frames = int(rng.integers(900, 2500))
        picks = np.arange(0, n_frames, 15)
        scores = rng.random(len(picks)).astype(np.float32)
        cuts = np.sort(rng.choice(np.arange(30, n_frames - 30), size=6, replace=False))
        starts = np.concatenate([[0], cuts, [n_frames + 10, n_frames + 90]])
        ends = np.concatenate([cuts - 1, [n_frames + 9, n_frames + 89, n_frames + 200 + 40 * j]])
        cps = np.stack([starts, ends], axis=1).astype(np.int64)
        summ = generate_summary([cps], [scores], [n_frames], [picks])[0]
        store.update({"v%d_scores" % j: scores, "v%d_picks" % j: picks, "v%d_cps" % j: cps,
                      "v%d_nframes" % j: n_frames, "v%d_summary" % j: summ})
        print("video", j, "n_frames", n_frames, "summary len", len(summ), "selected", int(summ.sum()))
    np.savez_compressed(os.path.join(HERE, "eval_nan_golden.npz"), **store)


if __name__ == "__main__":
    main()
