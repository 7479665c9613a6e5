#!/usr/bin/env python3
"""Random soak of the Linear entry points of the C ABI against float64: vs_linear_{f32,bf16,f16x3} (bias, optional ReLU,
optional positional rows) and vs_linear_residual_layernorm_{f32,bf16,f16x3} (+ score head, sigmoid) over random shapes -
M from 1 to a few thousand (ragged tiles, the latency kernels' and the tiled kernels' ranges: VS_SKINNY_ROWS pinned to 0 in
half of the cases), N and K multiples of 32 up to 1024 / 2048.  The bf16 entries are checked on bf16-rounded operands.

    python tools/fuzz_linear.py [seconds] [seed]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-summarization_amd")
lib = pkg._lib.load()
dev = torch.device("cuda:0")
MS = [1, 2, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 255, 256, 257, 320, 500, 777, 1024, 1280, 2049, 4100, 9000]
NS = [32, 64, 96, 128, 192, 256, 320, 512, 768, 1024]
KS = [32, 64, 128, 256, 320, 512, 1024, 2048]
TOL = {"f32": 3e-5, "f16x3": 3e-5, "bf16": 1e-4}          # relative to the largest output entry (bf16: on the rounded operands; measured 1-4e-6 everywhere)


def rnd16(t):
    return t.to(torch.bfloat16).double()


def main(budget, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    g = torch.Generator().manual_seed(seed)
    st = torch.cuda.current_stream().cuda_stream
    t_end, n, t_print = time.time() + budget, 0, time.time() + 30
    worst = {}
    try:
        while time.time() < t_end:
            prec = str(rng.choice(["f32", "bf16", "f16x3"]))
            ln = bool(rng.integers(2))
            M, K = int(rng.choice(MS)), int(rng.choice(KS))
            N = int(rng.choice([128, 256] if ln else NS))          # the LayerNorm GEMMs: d_model 128 / 256 (wider models take GEMM + row pass)
            if ln and prec == "bf16" and K % 64:
                continue
            pin = int(rng.integers(2))
            pkg._lib.set_option("VS_SKINNY_ROWS", 0 if pin else -1)
            pkg._lib.set_option("VS_LP_MIN_ROWS", 0)
            A = torch.randn(M, K, generator=g) * float(rng.choice([0.3, 1.0, 3.0]))
            W = torch.randn(N, K, generator=g) * (K ** -0.5)
            bias = torch.randn(N, generator=g)
            Ad, Wd, bd = A.to(dev), W.to(dev), bias.to(dev)
            A64, W64 = (rnd16(A), rnd16(W)) if prec == "bf16" else (A.double(), W.double())
            ref = A64 @ W64.t() + bias.double()
            tag = "%s%s M=%d N=%d K=%d pin=%d" % (prec, "+ln" if ln else "", M, N, K, pin)
            if not ln:
                relu = int(rng.integers(2))
                T = int(rng.integers(1, M + 1))
                pe = torch.randn(T, N, generator=g) if (rng.integers(2) and not relu) else None       # (the ABI: positional rows without ReLU)
                out = torch.full((M, N), float("nan"), device=dev)
                fn = {"f32": lib.vs_linear_f32, "bf16": lib.vs_linear_bf16, "f16x3": lib.vs_linear_f16x3}[prec]
                ped = None if pe is None else pe.to(dev)
                pkg._lib.check(fn(Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr(), out.data_ptr(), M, N, K, relu, None if ped is None else ped.data_ptr(), T, st))
                if relu:
                    ref = ref.clamp_min(0.0)
                if pe is not None:
                    ref = ref + pe.double()[torch.arange(M) % T]
                tag += " relu=%d pe=%d" % (relu, pe is not None)
                pairs = [("C", out, ref)]
            else:
                res = torch.randn(M, N, generator=g)
                gamma, beta = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.1
                nc = int(rng.choice([0, 1, 3]))
                sig = int(rng.integers(2))
                sw, sb = torch.randn(max(nc, 1), N, generator=g) * N ** -0.5, torch.randn(max(nc, 1), generator=g)
                out = torch.full((M, N), float("nan"), device=dev)
                scores = torch.full((M, max(nc, 1)), float("nan"), device=dev)
                rd, gd, btd, swd, sbd = res.to(dev), gamma.to(dev), beta.to(dev), sw.to(dev), sb.to(dev)
                fn = {"f32": lib.vs_linear_residual_layernorm_f32, "bf16": lib.vs_linear_residual_layernorm_bf16,
                      "f16x3": lib.vs_linear_residual_layernorm_f16x3}[prec]
                pkg._lib.check(fn(Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr(), rd.data_ptr(), gd.data_ptr(), btd.data_ptr(), out.data_ptr(), M, N, K,
                                  swd.data_ptr() if nc else None, sbd.data_ptr() if nc else None, nc, sig, scores.data_ptr() if nc else None, st))
                z = ref + res.double()
                y = (z - z.mean(dim=1, keepdim=True)) / torch.sqrt(z.var(dim=1, unbiased=False, keepdim=True) + 1e-5) * gamma.double() + beta.double()
                pairs = [("out", out, y)]
                if nc:
                    sc = y @ sw.double().t() + sb.double()
                    pairs.append(("scores", scores, torch.sigmoid(sc) if sig else sc))
                tag += " nc=%d sig=%d" % (nc, sig)
            torch.cuda.synchronize()
            for name, got, want in pairs:
                gotd = got.cpu().double()
                assert bool(torch.isfinite(gotd).all()), "non-finite %s: %s" % (name, tag)
                err = (gotd - want).abs().max().item() / (want.abs().max().item() + 1e-30)
                key = prec + ("+ln" if ln else "")
                worst[key] = max(worst.get(key, 0.0), err)
                assert err < TOL[prec], "%s rel err %.3e: %s" % (name, err, tag)
            n += 1
            if time.time() > t_print:
                print("  ... %d cases, worst %s" % (n, ", ".join("%s %.2e" % kv for kv in sorted(worst.items()))), flush=True)
                t_print = time.time() + 30
    finally:
        pkg._lib.set_option("VS_SKINNY_ROWS", -1)
        pkg._lib.set_option("VS_LP_MIN_ROWS", -1)
    print("fuzz_linear: %d cases clean in %.0f s (seed %d); worst error relative to the largest output entry: %s (bounds %s)" % (
        n, budget, seed, ", ".join("%s %.2e" % kv for kv in sorted(worst.items())), TOL))


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
