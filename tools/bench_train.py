#!/usr/bin/env python3
"""Training step timing: forward under autograd + loss + backward on the HIP training path, at the
reference's own regime (train.py: batch 4, a few hundred frames) and at the bench shape (B=64, T=1024), M-A.
Beside it, for context only, the same step composed from torch ops on the same GPU (tests/torch_ref.py, rocBLAS /
ATen kernels - what the reference's nn.Module does on a GPU, minus its per-layer .cpu() copy of the attention maps).

    python tools/bench_train.py [--dropout 0.3] [--torch]"""
import argparse
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("video-summarization_amd")

ap = argparse.ArgumentParser()
ap.add_argument("--dropout", type=float, default=0.3)
ap.add_argument("--torch", action="store_true", help="also time the composed-torch step (context)")
ap.add_argument("--shapes", default="4x320,4x640,16x1024,64x1024")
ap.add_argument("--model", default="A")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--only-bf16", action="store_true", help="time only the set_train_dtype('bf16') step (for a kernel profile)")
args = ap.parse_args()
dev = torch.device("cuda:0")
H, d, L = (4, 256, 4) if args.model == "A" else (4, 512, 3)
sd = pkg.synth.make_state_dict(d, L, 1234)
m = pkg.SimNet(num_heads=H, d_model=d, num_layers=L, sparsity=0.0, dropout=args.dropout)
m.load_state_dict(sd)
m = m.to(dev).train()


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for shape in args.shapes.split(","):
    B, T = (int(v) for v in shape.split("x"))
    x = torch.randn(B, T, 1024, device=dev)
    lengths = [T - (37 * i) % (T // 3) for i in range(B)]
    for b, n in enumerate(lengths):
        x[b, n:] = 1000.0
    mask = x[:, :, 0] == 1000
    target = torch.rand(B, T, device=dev)

    def step():
        pred, _ = m(x, mask)
        loss = pkg.mse_with_mask_loss(pred, target, mask)
        m.zero_grad(set_to_none=True)
        loss.backward()

    # the reference's whole train_step (train.py:111-131): autocast forward, masked MSE, GradScaler, Adam - every step
    # re-packs the parameters the optimizer wrote (vs_weights_update) and rebuilds the dgrad transposes
    optim = torch.optim.Adam(m.parameters(), lr=1e-5, weight_decay=1e-5)
    scaler = torch.amp.GradScaler("cuda")

    def full_step():
        with torch.autocast("cuda"):
            pred, _ = m(x, mask)
            loss = pkg.mse_with_mask_loss(pred, target, mask)
        optim.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        scaler.step(optim)
        scaler.update()

    def fwd_only():
        pred, _ = m(x, mask)
        return pred

    if args.only_bf16:
        m.set_train_dtype("bf16")
        print("B=%3d T=%4d  bf16 fwd+loss+bwd %.3f ms" % (B, T, timed(step, args.iters)), flush=True)
        m.set_train_dtype("fp32")
        continue
    with torch.no_grad():
        ev = timed(lambda: m.eval()(x, mask), args.iters)
    m.train()
    f = timed(fwd_only, args.iters)
    s = timed(step, args.iters)
    fs = timed(full_step, args.iters)
    m.set_train_dtype("bf16")           # the autocast counterpart: bf16 Linear / dgrad / wgrad GEMMs
    s16 = timed(step, args.iters)
    fs16 = timed(full_step, args.iters)
    m.set_train_dtype("fp16")           # ... and with the reference's own 16-bit type (tiled GEMMs everywhere, no A-stationary form)
    sh16 = timed(step, args.iters)
    fsh16 = timed(full_step, args.iters)
    m.set_train_dtype("fp32")
    flops_f = B * T * (2 * 1024 * d + L * (24 * d * d + 4 * T * d))
    # backward: 2x the Linear flops (dgrad + wgrad) + 3.5x the attention flops (7 products for the forward's 2)
    flops_b = B * T * (2 * 2 * 1024 * d + L * (2 * 24 * d * d + 14 * T * d)) - B * T * 2 * 1024 * d   # no input gradient
    line = "B=%3d T=%4d  scoring fwd %.3f ms | train fwd %.3f ms | fwd+loss+bwd %.3f ms (bwd %.3f ms) | %.1f TF fwd, %.1f TF bwd, %.0f frames/s | WHOLE train_step (autocast + GradScaler + Adam + re-pack) %.3f ms = %.0f frames/s trained" % (
        B, T, ev, f, s, s - f, flops_f / f / 1e9, flops_b / (s - f) / 1e9, B * T / s * 1e3, fs, B * T / fs * 1e3)
    line += " | bf16 GEMMs (set_train_dtype): fwd+loss+bwd %.3f ms, whole train_step %.3f ms = %.0f frames/s" % (s16, fs16, B * T / fs16 * 1e3)
    line += " | fp16 GEMMs: fwd+loss+bwd %.3f ms, whole train_step %.3f ms = %.0f frames/s" % (sh16, fsh16, B * T / fsh16 * 1e3)
    if args.torch:
        import torch_ref
        params = {k: v.to(dev).clone().requires_grad_(v.dtype.is_floating_point and "pos_embedding" not in k) for k, v in sd.items()}

        def tstep():
            pred, _ = torch_ref.forward_with_masks(params, x, mask, H)
            loss = (((pred.squeeze(2) - target) * (~mask).float()) ** 2).mean()
            for p_ in params.values():
                p_.grad = None
            loss.backward()
        line += " | composed torch (no dropout) %.3f ms" % timed(tstep, max(2, args.iters // 2))
    print(line, flush=True)
