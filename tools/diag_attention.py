#!/usr/bin/env python3
"""Diagnostic: where does a block of the exact attention kernel spend its cycles, and what separates
T = 1024 (0.82 of the fp32 MFMA peak) from T = 8192 (0.87)?  Runs the STAMPED instantiation of attn_fwd_pipe<64, 8 waves>
(diagnostic library only: -DVS_WITH_DIAG) and prints, per wave, the cycles of every phase of a 64-key tile, the fixed
cost per block (prologue + epilogue) and the spread of block start / end times.  Never part of a product run.

    python tools/diag_attention.py [B H T]..."""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-summarization_amd")
os.environ["VS_LIBRARY"] = pkg._lib.build(diag=True) if not os.path.exists(pkg._lib.DIAG_LIB_PATH) else pkg._lib.DIAG_LIB_PATH
lib = pkg._lib.load()
lib.vs_diag_attention.restype = C.c_int
lib.vs_diag_attention.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 3 + [C.c_float, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
NAMES = ["prologue(K/V tile 0, S'00)", "barrier A", "S'(t,1) || softmax(t,0) + staging", "P.V (t,0)", "barrier B",
         "S'(t+1,0) || softmax(t,1)", "P.V (t,1)", "epilogue (O/l, stores)", "total", "", "", "", "Q fetch"]


def run(B, H, T, dh=64):
    q, k, v = (torch.randn(B, H, T, dh, device=dev) for _ in range(3))
    out = torch.empty(B, T, H * dh, device=dev)
    nblk = 8 * ((B * H + 7) // 8) * ((T + 255) // 256)
    diag = torch.zeros(nblk * 8 * 16, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    scale = (H * dh) ** -0.5
    warm()
    for _ in range(3):
        rc = lib.vs_diag_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, T, scale, diag.data_ptr(), st)
        assert rc == nblk, rc
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.vs_diag_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, T, scale, diag.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 4.0 * B * H * T * T * dh
    # the product kernel on the same inputs, for the instrumentation overhead
    lib.vs_attention_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(), B, H, T, dh, scale, st)
    e0.record()
    for _ in range(10):
        lib.vs_attention_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(), B, H, T, dh, scale, st)
    e1.record()
    torch.cuda.synchronize()
    ms_p = e0.elapsed_time(e1) / 10
    d = diag.view(nblk * 8, 16).double().cpu()
    d = d[d[:, 8] > 0]
    nt = d[:, 11].median().item()
    tot = d[:, 8]
    print("B=%d H=%d T=%d: stamped %.4f ms (%.1f TF), product kernel %.4f ms (%.1f TF = %.3f of 157.3); %d blocks x 8 waves, %d tiles per block"
          % (B, H, T, ms, fl / ms / 1e9, ms_p, fl / ms_p / 1e9, fl / ms_p / 1e9 / 157.3, nblk, nt))
    mf_tile = 130.0                      # MFMAs per wave and 64-key tile (2 x (1 bias + 32 S' + 32 P.V))
    print("   cycles per wave: total med %.0f (min %.0f max %.0f); ideal = 2 waves/SIMD x %d tiles x %d MFMAs x 64 = %.0f -> MFMA share %.3f"
          % (tot.median().item(), tot.min().item(), tot.max().item(), nt, mf_tile, 2 * nt * mf_tile * 64, 2 * nt * mf_tile * 64 / tot.median().item()))
    for i in (12, 0, 1, 2, 3, 4, 5, 6, 7):
        c = d[:, i]
        per = c / (nt if 1 <= i <= 6 else 1)
        print("   %-36s %6.2f %% of the wave | %8.0f cycles %s (min %.0f max %.0f)" % (
            NAMES[i], 100 * c.mean().item() / tot.mean().item(), per.median().item(), "per tile" if 1 <= i <= 6 else "per block",
            per.min().item(), per.max().item()))
    fixed = (d[:, 0] + d[:, 12] + d[:, 7]).median().item()
    loop = (d[:, 1:7].sum(1)).median().item()
    print("   fixed per block (Q fetch + prologue + epilogue) %.0f cycles = %.2f %% ; tile loop %.0f cycles = %.0f per tile (2 x 65 MFMAs x 64 x 2 waves = %.0f)"
          % (fixed, 100 * fixed / tot.median().item(), loop, loop / nt, 2 * 65 * 64 * 2))
    t0, t1 = d[:, 9], d[:, 10]
    base = t0.min()
    life = (t1 - t0) / 100
    print("   wall: wave start med %.1f max %.1f us | wave end min %.1f med %.1f max %.1f us | kernel %.1f us | wave lifetime med %.1f us -> in-wave clock %.2f GHz"
          % (((t0 - base).median() / 100).item(), ((t0 - base).max() / 100).item(), ((t1 - base).min() / 100).item(),
             ((t1 - base).median() / 100).item(), ((t1 - base).max() / 100).item(), ms * 1e3, life.median().item(),
             tot.median().item() / life.median().item() / 1e3))
    # rounds: blocks per CU = nblk / 256; idle gaps between rounds show as (kernel time - rounds x lifetime)
    rounds = nblk / 256.0
    print("   %.1f block rounds per CU x lifetime %.1f us = %.1f us vs kernel %.1f us -> %.1f %% of the launch outside any wave's lifetime (ramp, tail, dispatch gaps)"
          % (rounds, life.median().item(), rounds * life.median().item(), ms * 1e3, 100 * (1 - rounds * life.median().item() / (ms * 1e3))))


def warm(seconds=0.4):
    """the chip raises its clock over the first few hundred ms of load: keep it busy before a measurement"""
    import time
    a = torch.randn(4096, 4096, device=dev)
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(10):
            a = (a @ a).clamp_(-1, 1)
        torch.cuda.synchronize()


LP_NAMES = {0: "prologue", 1: "phase A: S(t+1) MFMAs || softmax(t)", 2: "phase B: P.V + row sums || staging", 3: "barrier", 7: "epilogue"}
lib.vs_diag_attention_lp.restype = C.c_int
lib.vs_diag_attention_lp.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 3 + [C.c_float, C.c_int32, C.c_void_p, C.c_void_p]


def run_lp(B, H, T, prec, dh=64):
    """the low-precision kernels: prec 1 = bf16 in / out (the bf16 mode's form), 2 = fp16x3 on fp32 tensors"""
    dt = torch.bfloat16 if prec == 1 else torch.float32
    q, k, v = (torch.randn(B, H, T, dh, device=dev) for _ in range(3))
    if prec == 1:
        q = q * ((H * dh) ** -0.5 * 1.4426950408889634)               # stored pre-scaled in the bf16 mode
    q, k, v = q.to(dt), k.to(dt), v.to(dt)
    out = torch.empty(B, T, H * dh, device=dev, dtype=dt)
    nblk = 8 * ((B * H + 7) // 8) * ((T + 255) // 256)
    diag = torch.zeros(nblk * 8 * 16, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    scale = (H * dh) ** -0.5
    warm()
    for _ in range(3):
        rc = lib.vs_diag_attention_lp(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, T, scale, prec, diag.data_ptr(), st)
        assert rc == nblk, rc
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.vs_diag_attention_lp(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, T, scale, prec, diag.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 4.0 * B * H * T * T * dh
    d = diag.view(nblk * 8, 16).double().cpu()
    d = d[d[:, 8] > 0]
    nt = d[:, 11].median().item()
    tot = d[:, 8]
    nm = 20 if prec == 1 else 56
    print("%s B=%d H=%d T=%d: stamped %.4f ms (%.1f TF = %.3f of 2500 / %d products); %d tiles per block"
          % ("bf16" if prec == 1 else "fp16x3", B, H, T, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500 * (1 if prec == 1 else 3), 1 if prec == 1 else 3, nt))
    print("   cycles per wave: total med %.0f; per tile %.0f; MFMA pipe per tile = 2 waves x %d MFMAs x 32 = %d -> MFMA share %.3f"
          % (tot.median().item(), tot.median().item() / nt, nm, 2 * nm * 32, 2 * nm * 32 * nt / tot.median().item()))
    for i in (0, 1, 2, 3, 7):
        c = d[:, i]
        per = c / (nt if i in (1, 2, 3) else 1)
        print("   %-40s %6.2f %% of the wave | %8.0f cycles %s (min %.0f max %.0f)" % (
            LP_NAMES[i], 100 * c.mean().item() / tot.mean().item(), per.median().item(), "per tile" if i in (1, 2, 3) else "per block",
            per.min().item(), per.max().item()))
    life = (d[:, 10] - d[:, 9]) / 100
    print("   wave lifetime med %.1f us -> in-wave clock %.2f GHz; kernel %.1f us" % (life.median().item(), tot.median().item() / life.median().item() / 1e3, ms * 1e3))


def run_abl(B, H, T, dh=64):
    """timing-only ablations of the bf16 kernel (results are wrong by construction)"""
    q, k, v = (torch.randn(B, H, T, dh, device=dev) for _ in range(3))
    scale = (H * dh) ** -0.5
    q = (q * (scale * 1.4426950408889634)).to(torch.bfloat16)        # the bf16 mode's QKV epilogue stores q pre-scaled
    k, v = k.to(torch.bfloat16), v.to(torch.bfloat16)
    out = torch.empty(B, T, H * dh, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    fl = 4.0 * B * H * T * T * dh
    names = {0: "full kernel", 1: "no staging", 2: "no barriers", 3: "no staging, no barriers", 4: "fragment reads pinned to tile 0",
             5: "no staging, pinned reads", 7: "no staging, no barriers, pinned reads", 16: "no fragment reads (one fragment for all)",
             17: "no staging, no fragment reads", 19: "no staging, no barriers, no fragment reads"}
    warm()
    for abl in (0, 1, 2, 3, 4, 5, 7, 16, 17, 19):
        for _ in range(3):
            rc = lib.vs_diag_attention_lp(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, T, scale, 100 + abl, None, st)
            assert rc > 0, rc
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.vs_diag_attention_lp(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, T, scale, 100 + abl, None, st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("bf16 B=%d H=%d T=%d  ABL %2d %-44s %.4f ms  %.0f TF" % (B, H, T, abl, names[abl], ms, fl / ms / 1e9))


if len(sys.argv) > 1 and sys.argv[1] == "abl":
    run_abl(8, 4, 8192)
    run_abl(64, 4, 1024)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "lp":
    for prec in (1, 2):
        run_lp(8, 4, 8192, prec)
        run_lp(64, 4, 1024, prec)
    sys.exit(0)
shapes = [(8, 4, 8192), (64, 4, 1024)]
if len(sys.argv) > 3:
    a = [int(v) for v in sys.argv[1:]]
    shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
for s in shapes:
    run(*s)
