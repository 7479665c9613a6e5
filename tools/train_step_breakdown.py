#!/usr/bin/env python3
"""Where a reference-sized training step (B=4, T=320: train.py:111-131 - autocast forward, masked MSE, GradScaler, Adam) spends
its wall time: each phase timed with a device synchronisation behind it (so host enqueue AND device time are inside), then the
whole step without the inner synchronisations."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-summarization_amd")
dev = torch.device("cuda:0")
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 320)
m = pkg.SimNet(num_heads=4, d_model=256, num_layers=4, sparsity=0.0, dropout=0.3)
m.load_state_dict(pkg.synth.make_state_dict(256, 4, 1234)); m = m.to(dev).train()
x = torch.randn(B, T, 1024, device=dev); mask = torch.zeros(B, T, dtype=torch.bool, device=dev); target = torch.rand(B, T, device=dev)
optim = torch.optim.Adam(m.parameters(), lr=1e-5, weight_decay=1e-5)
scaler = torch.amp.GradScaler("cuda")
phases = {}


def tick(name, t0):
    torch.cuda.synchronize()
    phases[name] = phases.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()


def step(sync):
    t = time.perf_counter()
    with torch.autocast("cuda"):
        pred, _ = m(x, mask)
        loss = pkg.mse_with_mask_loss(pred, target, mask)
    if sync: t = tick("forward + loss (incl. re-pack of the parameters Adam wrote)", t)
    optim.zero_grad(set_to_none=True)
    scaler.scale(loss).backward()
    if sync: t = tick("backward", t)
    scaler.step(optim)
    if sync: t = tick("scaler.step (unscale + inf check + Adam)", t)
    scaler.update()
    if sync: t = tick("scaler.update", t)


for _ in range(10): step(False)
torch.cuda.synchronize()
N = 50
for _ in range(N): step(True)
t0 = time.perf_counter()
for _ in range(N): step(False)
torch.cuda.synchronize()
whole = (time.perf_counter() - t0) / N
print("B=%d T=%d: whole train_step %.3f ms (no inner synchronisation)" % (B, T, whole * 1e3))
for k, v in phases.items(): print("   %-70s %.3f ms" % (k, v / N * 1e3))
# host-only cost of our own glue: the packed-weights key + re-pack, and the backward's gradient views
t0 = time.perf_counter()
for _ in range(200): m._packed_weights(dev)
print("   _packed_weights() when nothing changed (key of 70 (data_ptr, version) pairs): %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))

if "--profile" in sys.argv:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(20): step(False)
        torch.cuda.synchronize()
    ev = prof.key_averages()
    tot = sum(e.self_device_time_total for e in ev)
    print("device time per step: %.3f ms" % (tot / 20 / 1e3))
    for e in sorted(ev, key=lambda e: -e.self_device_time_total)[:25]:
        if e.self_device_time_total > 0:
            print("   %-80s n/step %5.1f  %8.1f us/step" % (e.key[:80], e.count / 20, e.self_device_time_total / 20))
