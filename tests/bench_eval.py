#!/usr/bin/env python3
"""Times the C++ keyshot evaluation against the numpy/python oracle (the reference's algorithm) on the
golden 5-video set.  CPU only."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-summarization_amd")
ev = importlib.import_module("video-summarization_amd.evaluation")
from oracle import eval_oracle
G = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "eval_golden.npz"))
vids = [dict(scores=G["v%d_scores" % i], picks=G["v%d_picks" % i], cps=G["v%d_cps" % i], n=int(G["v%d_nframes" % i]),
             us=G["v%d_user_summary" % i], usc=G["v%d_user_scores" % i]) for i in range(5)]
def run_cpp():
    for v in vids:
        s = ev.generate_summary([v["cps"]], [v["scores"]], [v["n"]], [v["picks"]])[0]
        ev.evaluate_summary(s, v["us"], "avg"); ev.evaluate_scores(ev.upsample(v["scores"], v["n"], v["picks"]), v["usc"])
def run_oracle():
    for v in vids:
        s = eval_oracle.generate_summary(v["cps"], v["scores"], v["n"], v["picks"])
        eval_oracle.fscore(s, v["us"]); eval_oracle.rank_correlation(eval_oracle.upsample(v["scores"], v["n"], v["picks"]), v["usc"])
for name, fn, it in (("C++ (libvsscore)", run_cpp, 20), ("python/numpy/scipy oracle", run_oracle, 3)):
    fn(); t0 = time.perf_counter()
    for _ in range(it): fn()
    print("%-28s %8.2f ms per 5-video evaluation" % (name, (time.perf_counter() - t0) / it * 1e3))
