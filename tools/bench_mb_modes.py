"""M-B (H4 d512 L3: head dim 128) at B=64, T=1024: exact, bf16 attention, fp16x3 Linears, both.  GPU box."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-summarization_amd")
dev = torch.device("cuda:0")
m = pkg.SimNet(num_heads=4, d_model=512, num_layers=3, sparsity=0.0, dropout=0.3)
m.load_state_dict(pkg.synth.make_state_dict(512, 3, 5))
m = m.to(dev).eval()
x = torch.randn(64, 1024, 1024, device=dev)
with torch.no_grad():
    ref = m.score(x).clone()
    for lin, att in (("fp32", "fp32"), ("fp32", "bf16"), ("fp16x3", "fp32"), ("fp16x3", "bf16"), ("bf16", "bf16")):
        m.linear_dtype, m.attention_dtype = lin, att
        for _ in range(3): m.score(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): s = m.score(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print("M-B B=64 T=1024  linear %-6s attention %-5s: %7.3f ms  %.2f M frames/s  max|score - exact| %.2e" % (lin, att, dt * 1e3, 65536 / dt / 1e6, (s - ref).abs().max().item()))
