"""M-B (H4 d512 L3) bf16 / bf16 scoring at B=64, T=1024 only - the command tools' rocprofv3 kernel trace runs."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-summarization_amd")
dev = torch.device("cuda:0")
d, H, L = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 4, 3)))
m = pkg.SimNet(num_heads=H, d_model=d, num_layers=L, sparsity=0.0, dropout=0.3)
m.load_state_dict(pkg.synth.make_state_dict(d, L, 5))
m = m.to(dev).eval().set_compute_dtype("bf16")
x = torch.randn(64, 1024, 1024, device=dev)
with torch.no_grad():
    for _ in range(3): m.score(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m.score(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("d%d H%d L%d B=64 T=1024 bf16/bf16: %.3f ms" % (d, H, L, dt * 1e3))
