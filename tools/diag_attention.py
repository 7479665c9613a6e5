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


shapes = [(64, 4, 1024), (8, 4, 8192)]
if len(sys.argv) > 3:
    a = [int(v) for v in sys.argv[1:]]
    shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
for s in shapes:
    run(*s)
