#!/usr/bin/env python3
"""Times single stages of the scorer at bench shapes through the per-kernel C entry points.
usage: bench_stage.py attention|fc1|qkv|outproj|fc2 [iters]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-summarization_amd")
lib = pkg._lib.load()
dev = torch.device("cuda:0")
B, T, d, H = 64, 1024, 256, 4
M = B * T
st = torch.cuda.current_stream().cuda_stream
what = sys.argv[1] if len(sys.argv) > 1 else "attention"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def timeit(fn, flops):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("%s: %.4f ms  %.1f TFLOP/s  (%.1f%% of 157.3)" % (what, ms, flops / ms / 1e9, flops / ms / 1e9 / 1.573))


if what == "attention":
    q, k, v = (torch.randn(B, H, T, d // H, device=dev) for _ in range(3))
    out = torch.empty(B, T, d, device=dev)
    timeit(lambda: pkg._lib.check(lib.vs_attention_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(),
                                                       B, H, T, d // H, d ** -0.5, st)), 4.0 * M * T * d)
else:
    N, K, relu = {"fc1": (4 * d, d, 1), "qkv": (3 * d, d, 0), "embed": (d, 1024, 0)}.get(what, (d, d, 0))
    if what in ("outproj", "fc2"):
        K = d if what == "outproj" else 4 * d
        A = torch.randn(M, K, device=dev); W = torch.randn(d, K, device=dev) / K ** .5
        b = torch.randn(d, device=dev); res = torch.randn(M, d, device=dev); g = torch.ones(d, device=dev)
        out = torch.empty(M, d, device=dev)
        timeit(lambda: pkg._lib.check(lib.vs_linear_residual_layernorm_f32(
            A.data_ptr(), W.data_ptr(), b.data_ptr(), res.data_ptr(), g.data_ptr(), b.data_ptr(), out.data_ptr(),
            M, d, K, None, None, 0, 0, None, st)), 2.0 * M * d * K)
    else:
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** .5
        b = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev)
        timeit(lambda: pkg._lib.check(lib.vs_linear_f32(A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(),
                                                        M, N, K, relu, None, 0, st)), 2.0 * M * N * K)
