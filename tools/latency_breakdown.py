#!/usr/bin/env python3
"""Where does a single-video forward (B=1, T=320) spend its time: GPU stages vs host overhead."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-summarization_amd")
lib = pkg._lib.load()
dev = torch.device("cuda:0")
m = pkg.SimNet(num_heads=4, d_model=256, num_layers=4, sparsity=0.0, dropout=0.3)
m.load_state_dict(pkg.synth.make_state_dict(256, 4, 1234)); m = m.to(dev).eval()
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 320
x = torch.randn(B, T, 1024, device=dev)
if "splitk" in sys.argv[3:]:          # the opt-in latency mode (VS_FLAG_SPLITK)
    m.set_latency_mode(True)
with torch.no_grad():
    for _ in range(10): m(x)
    torch.cuda.synchronize()
    lib.vs_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(100): m(x)
    host = (time.perf_counter() - t0) / 100
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 100
    st = pkg._lib.profile_collect(); lib.vs_profile_enable(0)
    gpu = sum(ms for ms, n in st.values()) / 100
    print("B=%d T=%d [%s]: wall %.1f us/forward, host enqueue %.1f us, sum of GPU stage times %.1f us (stage events add a few us each)" % (
        B, T, "latency mode" if m.latency_mode else "default kernels", wall * 1e6, host * 1e6, gpu * 1e3))
    for k, (ms, n) in st.items(): print("   %-14s %7.1f us per launch (%d launches/forward)" % (k, ms / n * 1e3, n // 100))
