#!/usr/bin/env python3
"""Secondary measurements (BASELINE.json configs 2-4): single-video latency, M-B model, ragged corpus.
Not the headline metric (bench.py); numbers go to DESIGN.md."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-summarization_amd")
corpus = importlib.import_module("video-summarization_amd.corpus")
dev = torch.device("cuda:0")
MODELS = {"A": (4, 256, 4), "B": (4, 512, 3)}


def model(tag):
    H, d, L = MODELS[tag]
    m = pkg.SimNet(num_heads=H, d_model=d, num_layers=L, sparsity=0.0, dropout=0.3)
    m.load_state_dict(pkg.synth.make_state_dict(d, L, 1234))
    return m.to(dev).eval()


def timed(fn, iters, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


with torch.no_grad():
    for tag in ("A", "B"):
        m = model(tag)
        H, d, L = MODELS[tag]
        for B, T, it in ((1, 320, 200), (1, 1024, 200), (8, 1024, 50), (64, 1024, 20)):
            x = torch.randn(B, T, 1024, device=dev)
            F = 2 * 1024 * d + L * (24 * d * d + 4 * T * d) + 2 * d
            for mode in ("fp32", "fp16x3") + (("bf16",) if tag == "A" else ()):       # (bf16 Linears: d_model <= 256)
                m.set_compute_dtype(mode)
                dt = timed(lambda: m(x), it)
                print("cfg M-%s %-6s B=%2d T=%4d: %8.3f ms/forward  %10.0f frames/s  %6.1f TFLOP/s(fp32-equivalent)" % (
                    tag, mode, B, T, dt * 1e3, B * T / dt, B * T / dt * F / 1e12))
            m.set_compute_dtype("fp32")
    # configs[4] shape in fp32: B=8 videos x T=8192 frames x 2048-d features (beyond the reference's envelope)
    H, d, L = MODELS["A"]
    m5 = pkg.SimNet(num_heads=H, d_model=d, num_layers=L, sparsity=0.0, dropout=0.3, in_features=2048, pe_len=8192)
    m5.load_state_dict(pkg.synth.make_state_dict(d, L, 1234, in_features=2048, max_len=8192))
    m5 = m5.to(dev).eval()
    x5 = torch.randn(8, 8192, 2048, device=dev)
    dt = timed(lambda: m5(x5), 5, 2)
    F5 = 2 * 2048 * d + L * (24 * d * d + 4 * 8192 * d) + 2 * d
    print("cfg5 shape (fp32) B=8 T=8192 D=2048: %8.3f ms/forward  %10.0f frames/s  %6.1f TFLOP/s" % (dt * 1e3, 8 * 8192 / dt, 8 * 8192 / dt * F5 / 1e12))
    del m5, x5
    # configs[3] shape: 50 + 25 ragged videos, key masks on, one GPU
    m = model("A")
    g = torch.Generator().manual_seed(7)
    lens = torch.randint(150, 650, (50,), generator=g).tolist() + torch.randint(100, 650, (25,), generator=g).tolist()
    vids = [torch.randn(t, 1024, generator=g).to(dev) for t in lens]
    for mode in ("fp32", "fp16x3", "bf16"):
        m.set_compute_dtype(mode)
        for mf in (16384, 65536):
            for packed in (False, True):
                fn = lambda: corpus.score_corpus(lambda x, mk: m.score(x, mk), vids, device=dev, max_frames=mf,
                                                 packed_fn=(lambda x, ln: m.score_packed(x, ln)) if packed else None)
                dt = timed(fn, 10, 2)
                print("corpus 75 ragged videos (%d frames) %-6s max_frames=%5d %-6s: %.2f ms  %.0f frames/s (incl. host batching + D2H of scores)" % (
                    sum(lens), mode, mf, "packed" if packed else "padded", dt * 1e3, sum(lens) / dt))
