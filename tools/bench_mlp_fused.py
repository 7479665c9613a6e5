#!/usr/bin/env python3
"""The fused bf16 MLP block kernel (vs_mlp_block_bf16: the layer-tail kernel without its out-projection and QKV parts)
alone: time per launch and TFLOP/s at M rows (default 65536),
and with VS_MLP_ABLS=1,2,.. (diagnostic library; bit 1 no weight staging, 2 no chunk barrier, 4 no fragment LDS
reads, 8 no chunk loop at all) the timing-only ablations that say where the time goes.  Usage: python tools/bench_mlp_fused.py [M]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-summarization_amd")
abls = [int(a) for a in os.environ.get("VS_MLP_ABLS", "").split(",") if a]
if abls:
    os.environ["VS_LIBRARY"] = pkg._lib.build(diag=True) if not os.path.exists(pkg._lib.DIAG_LIB_PATH) else pkg._lib.DIAG_LIB_PATH
lib = pkg._lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
m = pkg.SimNet(num_heads=4, d_model=256, num_layers=1, sparsity=0.0, dropout=0.3)
m.load_state_dict(pkg.synth.make_state_dict(256, 1, 5))
m = m.to(dev).eval()
packed = m._packed_weights(dev)
h = torch.randn(M, 256, device=dev)
out = torch.empty_like(h)
st = torch.cuda.current_stream().cuda_stream


def run(tag):
    for _ in range(3):
        pkg._lib.check(lib.vs_mlp_block_bf16(packed.handle, 0, h.data_ptr(), out.data_ptr(), M, 0, 0, None, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        pkg._lib.check(lib.vs_mlp_block_bf16(packed.handle, 0, h.data_ptr(), out.data_ptr(), M, 0, 0, None, st))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("%-28s M=%d: %7.1f us  %6.1f TFLOP/s" % (tag, M, us, 4.0 * M * 1024 * 256 / us / 1e6))


run("MLP block kernel")
for a in abls:
    pkg._lib.set_option("VS_MLP_ABL", a)
    run("ablation %d" % a)
