#!/usr/bin/env python3
"""BASELINE configs[4] (long-video stress): B=8 videos x T=8192 frames x 2048-d features, model M-A,
attention in fp32 (default) and on the bf16 matrix pipe (SimNet.attention_dtype = "bf16").
Prints per-stage HIP-event times.  Usage: python tools/bench_long.py [B] [T]"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-summarization_amd")
_lib = importlib.import_module("video-summarization_amd._lib")
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
H, d, L, D = 4, 256, 4, int(os.environ.get("VS_BENCH_D", "2048"))
m = pkg.SimNet(num_heads=H, d_model=d, num_layers=L, sparsity=0.0, dropout=0.3, in_features=D, pe_len=max(T, 2000))
m.load_state_dict(pkg.synth.make_state_dict(d, L, 1234, in_features=D, max_len=max(T, 2000)))
m = m.to(dev).eval()
x = torch.randn(B, T, D, device=dev)
F = 2 * D * d + L * (24 * d * d + 4 * T * d) + 2 * d
lib = _lib.load()
with torch.no_grad():
    outs = {}
    for mode in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("fp32", "bf16-attn", "bf16")):
        m.set_compute_dtype("fp32")
        if mode == "bf16-attn":
            m.attention_dtype = "bf16"
        elif mode == "bf16":
            m.set_compute_dtype("bf16")
        elif mode == "fp16x3-lin":
            m.linear_dtype = "fp16x3"
        elif mode == "fp16x3":
            m.set_compute_dtype("fp16x3")
        for _ in range(3):
            m.score(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        iters = 10
        for _ in range(iters):
            s = m.score(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        outs[mode] = s
        lib.vs_profile_enable(1)
        for _ in range(iters):
            m.score(x)
        prof = _lib.profile_collect()
        lib.vs_profile_enable(0)
        ms = {k: v[0] for k, v in prof.items()}
        stages = ", ".join("%s %.3f" % (k, v / iters) for k, v in ms.items())
        att_fl = L * 4.0 * T * d * B * T
        print("mode %-9s  B=%d T=%d D=%d: %8.3f ms/forward  %10.0f frames/s  %6.1f TFLOP/s(model)  attention %.1f TFLOP/s"
              % (mode, B, T, D, dt * 1e3, B * T / dt, B * T / dt * F / 1e12, att_fl / (ms["attention"] / iters * 1e-3) / 1e12))
        print("   stages (ms/forward): " + stages)
    for k in outs:
        if k != "fp32" and "fp32" in outs:
            print("max |score_%s - score_fp32| = %.3e" % (k, (outs[k] - outs["fp32"]).abs().max().item()))
