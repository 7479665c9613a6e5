#!/usr/bin/env python3
"""The wide models' bf16 Linear: vs_linear_bf16_operands (vs_gemm_ring.hip: bf16 operands by LDS-DMA) beside vs_linear_bf16
(gemm_nt_128: fp32 operands rounded on their way into LDS) on the M-B / d768 / d1024 layer shapes at 65 536 rows.  GPU box."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-summarization_amd")
lib = pkg._lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, N, K, relu, c16 in (("fc1   d512", 2048, 512, 1, 1), ("fc2   d512", 512, 2048, 0, 0), ("qkv   d512", 1536, 512, 0, 1),
                               ("out   d512", 512, 512, 0, 0), ("fc1  d1024", 4096, 1024, 1, 1), ("fc2  d1024", 1024, 4096, 0, 0),
                               ("fc1   d768", 3072, 768, 1, 1), ("out   d768", 768, 768, 0, 0)):
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    A16, W16 = A.to(torch.bfloat16), W.to(torch.bfloat16)
    C = torch.empty(M, N, device=dev)
    C16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if c16 else torch.float32)
    t_old = timed(lambda: pkg._lib.check(lib.vs_linear_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), C.data_ptr(), M, N, K, relu, None, 0, st)))
    t_new = timed(lambda: pkg._lib.check(lib.vs_linear_bf16_operands(A16.data_ptr(), W16.data_ptr(), b.data_ptr(), C16.data_ptr(), M, N, K, relu, c16, st)))
    fl = 2.0 * M * N * K
    by_new = M * K * 2 + N * K * 2 + M * N * (2 if c16 else 4)
    print("%s  M=%d N=%4d K=%4d  fp32 operands -> fp32 C: %7.1f us %6.0f TF | bf16 operands -> %s C: %7.1f us %6.0f TF  (%.0f MB: %.2f TB/s)" % (
        name, M, N, K, t_old, fl / t_old / 1e6, "bf16" if c16 else "fp32", t_new, fl / t_new / 1e6, by_new / 1e6, by_new / t_new / 1e6), flush=True)
