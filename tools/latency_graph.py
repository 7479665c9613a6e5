#!/usr/bin/env python3
"""One T=320 video per call: eager launches against a captured graph replay (torch.cuda.CUDAGraph = hipGraph), default kernels
and latency mode.  The forward only enqueues on the current stream (no allocation after warm-up except torch's caching
allocator, no synchronisation), so it is capturable as it is."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-summarization_amd")
dev = torch.device("cuda:0")
# model: M-A (4 heads, d_model 256, 4 layers); VS_LAT_MODEL=B: the reference's argparse default (4, 512, 3); =C: its constructor default (8, 512, 4)
H_, d_, L_ = {"A": (4, 256, 4), "B": (4, 512, 3), "C": (8, 512, 4)}[os.environ.get("VS_LAT_MODEL", "A")]
m = pkg.SimNet(num_heads=H_, d_model=d_, num_layers=L_, sparsity=0.0, dropout=0.3)
m.load_state_dict(pkg.synth.make_state_dict(d_, L_, 1234)); m = m.to(dev).eval()
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 320)
x = torch.randn(B, T, 1024, device=dev)


def timed(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    for mode in (False, True):
        m.set_latency_mode(mode)
        eager = timed(lambda: m(x))
        ref = m(x)[0].clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): m(x)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m(x)
        g.replay(); torch.cuda.synchronize()
        same = torch.equal(out[0], ref)
        graph = timed(g.replay)
        print("H=%d d=%d L=%d B=%d T=%d %-13s eager %.4f ms | graph replay %.4f ms | replay == eager bitwise: %s" % (H_, d_, L_, B, T, "latency mode" if mode else "default", eager, graph, same), flush=True)
