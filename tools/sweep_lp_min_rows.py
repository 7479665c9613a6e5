#!/usr/bin/env python3
"""Where should the low-precision TRAINING kernels take over from the exact ones (VS_TRAIN_LP_MIN_ROWS)?  Times forward + loss +
backward of M-A at a ladder of batch sizes with set_train_dtype("fp32") and ("bf16", threshold forced to 0) and prints the
ratio; the committed output (profiles/r04_lp_min_rows_sweep.txt) is what the library's default threshold is read from.

    python tools/sweep_lp_min_rows.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-summarization_amd")
dev = torch.device("cuda:0")


def step_ms(m, x, mask, target, iters):
    def one():
        pred, _ = m(x, mask)
        loss = ((torch.sigmoid(pred.squeeze(-1)) - target) ** 2).mean()
        loss.backward()
        for p in m.parameters():
            p.grad = None
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        one()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    m = pkg.SimNet(num_heads=4, d_model=256, num_layers=4, sparsity=0.0, dropout=0.3)
    m.load_state_dict(pkg.synth.make_state_dict(256, 4, 3))
    m = m.to(dev).train()
    print("M-A (H4 d256 L4), dropout 0.3, forward + MSE loss + backward, ms per step; low-precision kernels forced on (VS_TRAIN_LP_MIN_ROWS = 0)")
    print("%8s %8s %10s %10s %8s" % ("B x T", "frames", "fp32 ms", "bf16 ms", "ratio"))
    for B, T in [(1, 320), (2, 320), (4, 320), (8, 320), (2, 1024), (16, 320), (4, 1024), (6, 1024), (8, 1024), (12, 1024), (16, 1024), (32, 1024)]:
        x = torch.randn(B, T, 1024, device=dev)
        target = torch.rand(B, T, device=dev)
        iters = 30 if B * T <= 4096 else 10
        m.set_train_dtype("fp32")
        t32 = step_ms(m, x, None, target, iters)
        try:
            pkg._lib.set_option("VS_TRAIN_LP_MIN_ROWS", 0)
            m.set_train_dtype("bf16")
            t16 = step_ms(m, x, None, target, iters)
            assert m.last_train_dtype == "bf16"
        finally:
            pkg._lib.set_option("VS_TRAIN_LP_MIN_ROWS", -1)
        print("%8s %8d %10.3f %10.3f %8.2f" % ("%dx%d" % (B, T), B * T, t32, t16, t32 / t16), flush=True)


if __name__ == "__main__":
    main()
