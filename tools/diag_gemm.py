#!/usr/bin/env python3
"""Diagnostic: where does a k-tile iteration of the persistent projection GEMM spend its cycles?
Runs the fc1-shaped GEMM (M=65536, N=1024, K=256) in the stamped DIAG build and prints the shares.
Never part of a product run; see cdna guide §7 'In-kernel stamps'."""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-summarization_amd")
# the stamped instantiations live in the diagnostic build only (libvsscore_diag.so, -DVS_WITH_DIAG)
os.environ["VS_LIBRARY"] = pkg._lib.build(diag=True) if not os.path.exists(pkg._lib.DIAG_LIB_PATH) else pkg._lib.DIAG_LIB_PATH
lib = pkg._lib.load()
lib.vs_diag_gemm.restype = C.c_int
lib.vs_diag_gemm.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p, C.c_void_p]

M, N, K = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (65536, 1024, 256)))
dev = torch.device("cuda:0")
A = torch.randn(M, K, device=dev)
W = torch.randn(N, K, device=dev) / K ** 0.5
b = torch.randn(N, device=dev)
Cc = torch.empty(M, N, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run(grid, diag):
    mode = int(os.environ.get("VS_DIAG_MODE", "3"))
    mode = 3 if mode == 1 else mode
    g = grid if grid > 0 else 512
    dbuf = torch.zeros(g * 4 * 8, dtype=torch.int64, device=dev)
    dp = dbuf.data_ptr() if mode in (2, 3) else None       # modes 2/3 time the diagnostic build itself
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        pkg._lib.check(lib.vs_diag_gemm(A.data_ptr(), W.data_ptr(), b.data_ptr(), Cc.data_ptr(), M, N, K, grid, dp, st))
    e0.record()
    for _ in range(10):
        pkg._lib.check(lib.vs_diag_gemm(A.data_ptr(), W.data_ptr(), b.data_ptr(), Cc.data_ptr(), M, N, K, grid, dp, st))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("grid %4d: %.4f ms  %.1f TFLOP/s" % (grid, ms, 2.0 * M * N * K / ms / 1e9))
    if mode in (2, 3):
        d = dbuf.view(g * 4, 8).double().cpu()
        cyc = d[:, 5].median().item()
        mf = 2.0 * M * N * K / 4096 / (256 * 4)          # MFMAs per SIMD
        t0, t1 = d[:, 3], d[:, 4]
        base = t0.min()
        print("   mode %d: cycles/wave min %.0f med %.0f max %.0f" % (mode, d[:, 5].min().item(), cyc, d[:, 5].max().item()))
        print("   wall (100 MHz ticks -> us): wave start min %.1f med %.1f max %.1f | wave end min %.1f med %.1f max %.1f | kernel %.1f us" % (
            0.0, ((t0 - base).median() / 100).item(), ((t0 - base).max() / 100).item(),
            ((t1 - base).min() / 100).item(), ((t1 - base).median() / 100).item(), ((t1 - base).max() / 100).item(), ms * 1e3))
        ends = ((t1 - base) / 100)[::4]          # one wave per block
        hist = torch.histc(ends, bins=12, min=ends.min().item(), max=ends.max().item())
        print("   block end-time histogram (%.0f..%.0f us): %s" % (ends.min().item(), ends.max().item(), [int(x) for x in hist.tolist()]))
        # do blocks b and b+256 (likely co-resident) pair up as early/late?
        if g == 512:
            e = ends.view(2, 256)
            print("   blocks 0-255 end med %.1f us, blocks 256-511 end med %.1f us; |pair difference| med %.1f us" % (
                e[0].median().item(), e[1].median().item(), (e[0] - e[1]).abs().median().item()))
        life = ((t1 - t0) / 100)
        print("   wave lifetime us: min %.1f med %.1f max %.1f -> in-wave clock %.2f GHz; MFMA-busy share of lifetime at 2 waves/SIMD = %.3f" % (
            life.min().item(), life.median().item(), life.max().item(), cyc / life.median().item() / 1e3,
            64.0 * (2.0 * M * N * K / 4096 / (g * 4)) * (g // 256) / cyc))
        return
    if diag:
        g = grid if grid > 0 else 512
        d = torch.zeros(g * 4 * 8, dtype=torch.int64, device=dev)
        pkg._lib.check(lib.vs_diag_gemm(A.data_ptr(), W.data_ptr(), b.data_ptr(), Cc.data_ptr(), M, N, K, grid, d.data_ptr(), st))
        torch.cuda.synchronize()
        d = d.view(g * 4, 8).double().cpu()
        names = ["issue_loads", "ds_read+mfma", "epilogue", "vmcnt+ds_write", "barrier"]
        tot = d[:, 5].mean().item()
        kt = d[:, 6].mean().item()
        print("   stamped build: mean cycles/wave %.0f over %.0f k-tiles = %.0f cycles per k-tile (64 MFMAs = 4096 busy)" % (tot, kt, tot / kt))
        for i, n in enumerate(names):
            print("   %-16s %6.1f %%   %7.0f cycles/k-tile   (min %.0f max %.0f per wave)" % (
                n, 100 * d[:, i].mean().item() / tot, d[:, i].mean().item() / kt, (d[:, i] / d[:, 6]).min().item(), (d[:, i] / d[:, 6]).max().item()))
        tb = d[:, 7]
        print("   start skew across waves: %.0f cycles" % (tb.max() - tb.min()).item())


for grid in ((256,) if os.environ.get('VS_DIAG_NWM') == '4' else (512, 256)):
    run(grid, True)
