#!/usr/bin/env python3
"""Quick check + timing of the one-wave-per-SIMD bf16 attention (csrc/vs_attention_w64.hip) against the 8-wave kernel
(VS_ATTN_W64 = 0) and an fp64 reference that shares the rounded operands.

    python tools/check_attn_w64.py            # correctness cases, then timings at the bench shapes
    python tools/check_attn_w64.py time       # timings only"""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-summarization_amd")
if len(sys.argv) > 1 and sys.argv[1] in ("abl", "ablone"):      # timing ablations: the diagnostic library (tools/gen_attn_w64.py --abl ... first)
    os.environ["VS_LIBRARY"] = pkg._lib.DIAG_LIB_PATH if os.path.exists(pkg._lib.DIAG_LIB_PATH) else pkg._lib.build(diag=True)
lib = pkg._lib.load()
dev = torch.device("cuda:0")
L2E = 1.4426950408889634


def ref64(q16, k16, v16, mask):
    s2 = torch.matmul(q16.double(), k16.double().transpose(2, 3))
    if mask is not None:
        s2 = s2.masked_fill(mask[:, None, None, :], float("-inf"))
    p = torch.exp2(s2 - s2.max(dim=3, keepdim=True).values)
    o = torch.matmul(p, v16.double()) / p.sum(dim=3, keepdim=True)
    B, H, T, dh = q16.shape
    return o.permute(0, 2, 1, 3).reshape(B, T, H * dh)


def run(q16, k16, v16, mask, w64):
    B, H, T, dh = q16.shape
    pkg._lib.set_option("VS_ATTN_W64", 1 if w64 else 0)
    out = torch.full((B, T, H * dh), float("nan"), device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    pkg._lib.check(lib.vs_attention_bf16_stored(q16.data_ptr(), k16.data_ptr(), v16.data_ptr(),
                                                mask.data_ptr() if mask is not None else None, out.data_ptr(), B, H, T, dh, st))
    torch.cuda.synchronize()
    pkg._lib.set_option("VS_ATTN_W64", -1)
    return out


def case(B, H, T, masked=False, sigma=2.0, seed=0, spike=False):
    g = torch.Generator().manual_seed(seed + T)
    q, k, v = (torch.randn(B, H, T, 64, generator=g) * sigma for _ in range(3))
    if spike and T > 301:
        k[:, :, 300] = q.mean(dim=2) * 50.0 + 20.0
        k[:, :, 77] = -k[:, :, 300]
    scale = (H * 64) ** -0.5 if not spike else 1.0
    q16 = (q * (scale * L2E)).to(torch.bfloat16).to(dev)
    k16, v16 = k.to(torch.bfloat16).to(dev), v.to(torch.bfloat16).to(dev)
    mask = None
    if masked:
        mask = torch.rand(B, T, generator=g) < 0.3
        mask[:, 0] = False
        if T > 130:
            mask[0, 64:128] = True              # a fully masked tile
            mask[-1, T // 2:] = True            # suffix padding
        mask = mask.to(dev)
    ref = ref64(q16.cpu(), k16.cpu(), v16.cpu(), mask.cpu() if mask is not None else None)
    new = run(q16, k16, v16, mask, True).cpu().double()
    old = run(q16, k16, v16, mask, False).cpu().double()
    en = (new - ref).abs().max().item() / ref.abs().max().item()
    eo = (old - ref).abs().max().item() / ref.abs().max().item()
    ok = bool(torch.isfinite(new).all()) and en < 6e-3
    print("B=%d H=%d T=%5d masked=%d spike=%d: new rel err %.2e (old kernel %.2e) %s" % (B, H, T, masked, spike, en, eo, "ok" if ok else "FAIL"), flush=True)
    return ok


def warm(seconds=0.5):
    """the chip settles its clock over the first few hundred ms of load: keep it busy before a measurement"""
    import time
    a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(10):
            a = (a @ a).clamp_(-1, 1)
        torch.cuda.synchronize()


def timing(B, H, T, iters=200, rounds=3):
    g = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(B, H, T, 64, generator=g) for _ in range(3))
    scale = (H * 64) ** -0.5
    q16 = (q * (scale * L2E)).to(torch.bfloat16).to(dev)
    k16, v16 = k.to(torch.bfloat16).to(dev), v.to(torch.bfloat16).to(dev)
    out = torch.empty((B, T, H * 64), device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    fl = 4.0 * B * H * T * T * 64
    res = {}
    names = {0: "8-wave kernel", 1: "w64, optimistic pass first", 2: "w64, checked pass only"}
    warm()
    for rnd in range(rounds):
        for var in (0, 1, 2):
            pkg._lib.set_option("VS_ATTN_W64", 0 if var == 0 else 1)
            pkg._lib.set_option("VS_ATTN_W64_CHECKED", 1 if var == 2 else 0)
            for _ in range(5):
                lib.vs_attention_bf16_stored(q16.data_ptr(), k16.data_ptr(), v16.data_ptr(), None, out.data_ptr(), B, H, T, 64, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                lib.vs_attention_bf16_stored(q16.data_ptr(), k16.data_ptr(), v16.data_ptr(), None, out.data_ptr(), B, H, T, 64, st)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(var, []).append(e0.elapsed_time(e1) / iters)
    for var in (0, 1, 2):
        ms = min(res[var])
        print("B=%d H=%d T=%d %-28s: %.4f ms (rounds %s)  %.0f TF (%.3f of 2.5 PF)" % (B, H, T, names[var], ms, " ".join("%.4f" % x for x in res[var]), fl / ms / 1e9, fl / ms / 1e9 / 2500), flush=True)
    pkg._lib.set_option("VS_ATTN_W64", -1)
    pkg._lib.set_option("VS_ATTN_W64_CHECKED", -1)
    return res


def ablations(B, H, T, iters=50):
    """timing ablations of the stream (WRONG results by construction): what is left when a class of work is removed"""
    g = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(B, H, T, 64, generator=g) for _ in range(3))
    q16 = (q * ((H * 64) ** -0.5 * L2E)).to(torch.bfloat16).to(dev)
    k16, v16 = k.to(torch.bfloat16).to(dev), v.to(torch.bfloat16).to(dev)
    out = torch.empty((B, T, H * 64), device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    fl = 4.0 * B * H * T * T * 64
    names = {0: "full kernel", 2: "no LDS-DMA", 4: "no fragment reads", 8: "no barrier", 14: "no DMA, no reads, no barrier",
             15: "MFMAs only", 16: "exp2 -> mov", 32: "no packs, no OR chain", 64: "proportional distribution",
             30: "exp2 -> mov, no DMA / reads / barrier"}
    pkg._lib.set_option("VS_ATTN_W64", 1)
    names.update({128: "no key-bias checks", 256: "no OR test (chain kept)", 384: "no bias checks, no OR test",
                  398: "no checks, no DMA / reads / barrier"})
    names.update({1: "no softmax work", 512: "no softmax beside S' MFMAs", 1024: "no softmax beside P.V MFMAs"})
    for abl in (0, 1, 2, 4, 8, 14, 15, 512, 1024):
        pkg._lib.set_option("VS_ATTN_W64_ABL", abl)
        warm(0.3)
        for _ in range(5):
            lib.vs_attention_bf16_stored(q16.data_ptr(), k16.data_ptr(), v16.data_ptr(), None, out.data_ptr(), B, H, T, 64, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            lib.vs_attention_bf16_stored(q16.data_ptr(), k16.data_ptr(), v16.data_ptr(), None, out.data_ptr(), B, H, T, 64, st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print("B=%d H=%d T=%d abl %2d %-30s %.4f ms  %5.0f TF" % (B, H, T, abl, names[abl], ms, fl / ms / 1e9), flush=True)
    pkg._lib.set_option("VS_ATTN_W64_ABL", 0)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "ablone":      # one ablation, for a counter run: python tools/check_attn_w64.py ablone 2
        B, H, T = 8, 4, 8192
        g = torch.Generator().manual_seed(1)
        q, k, v = (torch.randn(B, H, T, 64, generator=g) for _ in range(3))
        q16 = (q * ((H * 64) ** -0.5 * L2E)).to(torch.bfloat16).to(dev)
        k16, v16 = k.to(torch.bfloat16).to(dev), v.to(torch.bfloat16).to(dev)
        out = torch.empty((B, T, H * 64), device=dev, dtype=torch.bfloat16)
        st = torch.cuda.current_stream().cuda_stream
        pkg._lib.set_option("VS_ATTN_W64_ABL", int(sys.argv[2]))
        warm(0.3)
        for _ in range(30):
            lib.vs_attention_bf16_stored(q16.data_ptr(), k16.data_ptr(), v16.data_ptr(), None, out.data_ptr(), B, H, T, 64, st)
        torch.cuda.synchronize()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "abl":
        ablations(8, 4, 8192)
        ablations(64, 4, 1024)
        sys.exit(0)
    ok = True
    if len(sys.argv) < 2 or sys.argv[1] != "time":
        for args in [(1, 1, 64), (1, 1, 256), (2, 4, 320), (1, 4, 1024), (1, 2, 65), (1, 1, 1), (1, 4, 31), (2, 2, 513), (1, 2, 200),
                     (1, 1, 2048)]:
            ok &= case(*args)
        for args in [(2, 4, 200), (2, 2, 513), (1, 1, 64), (3, 2, 1000)]:
            ok &= case(*args, masked=True)
        ok &= case(1, 4, 512, spike=True)
        ok &= case(2, 2, 700, masked=True, spike=True)
        ok &= case(1, 2, 640, sigma=8.0)
    timing(8, 4, 8192, iters=50)
    timing(64, 4, 1024)
    timing(16, 4, 1024)
    timing(4, 4, 320, iters=400)
    print("ALL OK" if ok else "FAILURES")
    sys.exit(0 if ok else 1)
