#!/usr/bin/env python3
"""Wider models of the reference's envelope (any d_model % num_heads == 0, simnet.py:10-13): d_model 768 (12 heads of 64)
and 1024 (8 heads of 128), 3 layers, B=64, T=1024 - exact fp32, fp16x3 Linears (+ bf16 attention) and the bf16 mode."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-summarization_amd")
dev = torch.device("cuda:0")
x = torch.randn(64, 1024, 1024, device=dev)
for d, H, L in ((768, 12, 3), (1024, 8, 3)):
    m = pkg.SimNet(num_heads=H, d_model=d, num_layers=L, sparsity=0.0, dropout=0.3)
    m.load_state_dict(pkg.synth.make_state_dict(d, L, 5))
    m = m.to(dev).eval()
    F = 2 * 1024 * d + L * (24 * d * d + 4 * 1024 * d) + 2 * d
    with torch.no_grad():
        ref = m.score(x).clone()
        for mode in ("fp32", "fp16x3", "bf16"):
            m.set_compute_dtype(mode)
            if mode == "fp16x3" and d // H == 128:
                m.attention_dtype = "bf16"
            for _ in range(3):
                m.score(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                s = m.score(x)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
            print("d_model %4d H %2d L %d B=64 T=1024  %-6s (attention %-5s): %7.3f ms  %.2f M frames/s  %.0f TFLOP/s(model)  max|score - exact| %.2e"
                  % (d, H, L, mode, m.attention_dtype, dt * 1e3, 65536 / dt / 1e6, 65536 / dt * F / 1e12, (s - ref).abs().max().item()))
