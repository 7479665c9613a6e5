#!/usr/bin/env python3
"""A/B of the gemm_ln_rows de-phasing (VS_LN_STAGGER_PCT): stage times of out-proj+LN (K=256) and fc2+LN (K=1024) at
the bench shape for several start offsets of the second block of each CU."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-summarization_amd")
lib = pkg._lib.load()
dev = torch.device("cuda:0")
m = pkg.SimNet(num_heads=4, d_model=256, num_layers=4, sparsity=0.0, dropout=0.3)
m.load_state_dict(pkg.synth.make_state_dict(256, 4, 1234))
m = m.to(dev).eval()
x = torch.randn(64, 1024, 1024, device=dev)
with torch.no_grad():
    for pct in [int(a) for a in (sys.argv[1:] or "0 15 25 35 50 65 80 100".split())]:
        pkg._lib.set_option("VS_LN_STAGGER_PCT", pct)
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        lib.vs_profile_enable(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            m(x)
        e1.record()
        torch.cuda.synchronize()
        st = pkg._lib.profile_collect()
        lib.vs_profile_enable(0)
        print("stagger %3d%%: forward %.4f ms | outproj_ln %.4f ms  fc2_ln %.4f ms" % (
            pct, e0.elapsed_time(e1) / 30, st["outproj_ln"][0] / st["outproj_ln"][1], st["fc2_ln_score"][0] / st["fc2_ln_score"][1]), flush=True)
