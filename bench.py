#!/usr/bin/env python3
"""Headline benchmark: frames/sec scored on synthetic [B, T=1024, D=1024] (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torchrun environment the script LAUNCHES ITS OWN RANKS: before anything touches the GPU it
starts ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
bench.py ...`` as a child process, relays its output and exits with its code (a child, never an exec).  Started
by torchrun itself (RANK / WORLD_SIZE set) it is one rank of that job.

One "step" = one scorer forward (reference ``SimNet.forward``, logits + hidden state) over one batch of B=64
videos x T=1024 frames x 1024-d features per GPU (BASELINE.json configs[2], model M-A = heads 4, d_model 256,
layers 4 — run_finetune.sh:1), inputs already resident in HBM.  Videos are independent, so N GPUs score N disjoint
batches (weak scaling) and RCCL only gathers the [B,T] score matrices (one async all_gather per step, overlapped
with the next step's kernels).

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline      the dominant kernel: algorithmic FLOPs per launch / its mean launch duration, from HIP events the
                library records around every launch in a SEPARATE profiled pass of the same K steps (the headline
                `value` is timed with profiling off); `traffic` = HBM bytes per launch from the committed PMC passes
                (profiles/r04_hbm_traffic.json, stamped with the kernel-source hash it was measured on; null when
                the loaded sources differ)
  cpu_baseline  the oracle (CPU restatement of the reference, "port") timed on this box's host cores:
                B=8,T=1024 (throughput leg, = `value`'s unit) and B=1,T=320 x20 (SURVEY §8(d) latency leg)
  latency       configs[1]: one T=320 video, GPU ms per forward (same model), beside the CPU's
  emulated_f32  the same K steps again with every product EMULATED on the f16 matrix pipe ("fp16x3"): reported
                beside `value`, never as `value`: the headline is the exact-fp32 MFMA path (--compute to change).
  bf16_mode     the same K steps on the bf16 matrix pipe (reduced precision, opt-in): context only.
  training_step one training step on the same batch (HIP forward with dropout + masked MSE + HIP backward; SURVEY §8(f)
                row 2), exact fp32 and under set_train_dtype("bf16"): context only.

``--workload`` selects the BASELINE.json configuration (same launcher, same JSON contract, one rank-0 line):
  batch  (default) configs[2]: B=64 x T=1024 x D=1024 per GPU, exact fp32, weak scaling - the headline metric
  corpus configs[3]: the 75-video TVSum+SumMe-shaped ragged corpus (30 568 frames), STRONG scaling: the videos are dealt
         to the ranks (corpus.plan_shards), each rank scores its shard as packed batches, one all_gather returns every
         video's scores to every rank; a step = one pass over the corpus; `eval_ms` = the sharded keyshot evaluation
         (four sums all-reduced) of the same scores, timed beside the step, never inside `value`
  long   configs[4]: B=8 x T=8192 x D=2048 per GPU, every product on the bf16 matrix pipe (the config names bf16), weak
         scaling; `roofline.peak` is the dense bf16 MFMA peak
"""
import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
PEAK_F16_MFMA_TFLOPS = 2500.0     # dense f16 / bf16 matrix peak; an fp16x3 product costs 3 f16 products
# HBM bytes per launch per kernel come from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; see
# tools/collect_traffic.sh), committed under profiles/: counters cannot be read from inside the timed run.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r04_hbm_traffic.json")
STAGE_KERNEL = {"attention": "attn_fwd_pipe", "embed_pe": "gemm_nt_128<2", "qkv_proj": "gemm_nt_128<3",
                "fc1_relu": "gemm_nt_128<1", "outproj_ln": "gemm_ln_rows", "fc2_ln_score": "gemm_ln_rows"}


FORWARD_SOURCES = ("vs_attention.hip", "vs_device.h", "vs_kernels.h", "vs_kernels.hip", "vs_mlp_fused.hip", "vs_scorer.cpp")


def kernel_source_hash():
    """sha256 (16 hex) over the sources of the scoring forward (the kernels the traffic file was measured on and
    their launch code): the stamp of the traffic file."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "video-summarization_amd", "csrc")
    for name in FORWARD_SOURCES:
        h.update(name.encode())
        h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(stage):
    """(HBM bytes per launch of the kernel behind `stage`, provenance) from the committed PMC passes; bytes are
    None when the file is missing or was measured on other kernel sources than the ones loaded now."""
    try:
        t = json.load(open(TRAFFIC_FILE))
    except (OSError, ValueError):
        return None, "no traffic file"
    meta = t.get("_meta", {})
    src = "%s @ csrc %s" % (os.path.relpath(TRAFFIC_FILE, ROOT), meta.get("csrc_sha256_16", "?"))
    if meta.get("csrc_sha256_16") != kernel_source_hash():
        return None, src + " (stale: loaded sources are %s)" % kernel_source_hash()
    for name, v in t.items():
        if name.startswith(STAGE_KERNEL.get(stage, "?")):
            return int(v["hbm_bytes_per_launch"]), src
    return None, src


def stage_flops(B, T, Din, d, H, L):
    """Algorithmic FLOPs per LAUNCH of each stage (per-frame figure x B*T frames)."""
    M = B * T
    return {
        "embed_pe": 2.0 * M * Din * d,
        "qkv_proj": 2.0 * M * d * 3 * d,
        "attention": 4.0 * M * T * d,            # QK^T + PV: 2*T*d each per frame
        "outproj_ln": 2.0 * M * d * d,
        "fc1_relu": 2.0 * M * d * 4 * d,
        "fc2_ln_score": 2.0 * M * 4 * d * d + 2.0 * M * d,
    }


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks as a CHILD job (nothing in this
    process has touched the GPU yet; it only waits and passes the child's exit code on)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["VS_BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.call(cmd, env=env)


def dry_run(rank, world):
    """VS_BENCH_DRYRUN=1 (CPU test of the launch plumbing only, tests/test_bench_launch.py): every rank joins a gloo
    group, the ranks all_gather their rank ids, rank 0 prints one JSON line.  No scorer, no GPU, no numbers."""
    import torch
    import torch.distributed as dist
    ids = [rank]
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        got = torch.empty(world, dtype=torch.int64)
        dist.all_gather_into_tensor(got, torch.tensor([rank], dtype=torch.int64))
        ids = got.tolist()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dryrun": True, "n_gpus": world, "ranks": ids,
                          "self_launched": os.environ.get("VS_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
    return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="batch", choices=["batch", "corpus", "long"],
                    help="batch: configs[2] (headline); corpus: configs[3] (strong scaling); long: configs[4] (bf16)")
    ap.add_argument("--batch", type=int, default=None, help="videos per GPU per step (batch: 64, long: 8)")
    ap.add_argument("--frames", type=int, default=None, help="frames per video (batch: 1024, long: 8192)")
    ap.add_argument("--model", default="A", choices=["A", "B"], help="A: H4 d256 L4 (run scripts); B: H4 d512 L3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--compute", default=None, choices=["fp32", "fp16x3", "bf16"],
                    help="matrix-product arithmetic of the timed path (default: exact fp32 MFMA; workload long: bf16)")
    ap.add_argument("--no-emulated", action="store_true", help="skip the secondary fp16x3 measurement")
    ap.add_argument("--no-extras", action="store_true", help="skip the PCIe-inclusive and latency legs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline sample length")
    args = ap.parse_args()
    long_ = args.workload == "long"
    args.batch = args.batch or (8 if long_ else 64)
    args.frames = args.frames or (8192 if long_ else 1024)
    args.compute = args.compute or ("bf16" if long_ else "fp32")
    return args


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))          # before any GPU call, before torch is even imported
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    if os.environ.get("VS_BENCH_DRYRUN") == "1":
        return dry_run(rank, world)
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU path for the scorer)"
    # VS_BENCH_REHEARSE=1: rehearsal of the N > 1 control flow on a ONE-GPU box (every rank on cuda:0, gloo instead
    # of RCCL, which refuses two ranks on one device).  Its numbers mean nothing; it only proves the path runs.
    rehearse = os.environ.get("VS_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    backend = None
    # VS_BENCH_FORCE_DIST=1: join a process group even with one rank, so that a ONE-GPU box runs the real RCCL
    # all_gather of the N-GPU path (tests/test_bench_launch.py); the numbers are those of N = 1.
    force_dist = os.environ.get("VS_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            assert torch.cuda.device_count() >= world, "one GPU per rank: %d ranks, %d devices" % (world, torch.cuda.device_count())
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        backend = dist.get_backend()
        assert dist.get_world_size() == world and dist.get_rank() == rank
        assert rehearse or backend == "nccl", "multi-GPU runs gather over RCCL (torch backend 'nccl'), got %r" % backend
    ranks_seen = [0]
    if dist is not None:      # every rank's id through the collective transport itself (RCCL on a GPU node): all present, once each
        ids = torch.empty(world, dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        dist.all_gather_into_tensor(ids, torch.tensor([rank], dtype=torch.int64, device=ids.device))
        ranks_seen = ids.cpu().tolist()
        assert sorted(ranks_seen) == list(range(world)), ranks_seen

    pkg = importlib.import_module("video-summarization_amd")
    lib = pkg._lib.load()
    H, d, L = (4, 256, 4) if args.model == "A" else (4, 512, 3)
    wl = args.workload
    B, T, Din = args.batch, args.frames, (2048 if wl == "long" else 1024)
    if wl == "long":       # configs[4]: outside the reference's envelope (in_features 1024, 2000-row table): re-parameterised
        sd = pkg.synth.make_state_dict(d, L, seed=1234, in_features=Din, max_len=max(T, 2000))
        model = pkg.SimNet(num_heads=H, d_model=d, num_layers=L, sparsity=0.0, dropout=0.3, in_features=Din, pe_len=max(T, 2000))
    else:
        sd = pkg.synth.make_state_dict(d, L, seed=1234)
        model = pkg.SimNet(num_heads=H, d_model=d, num_layers=L, sparsity=0.0, dropout=0.3)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval().set_compute_dtype(args.compute)
    corpus_mod = importlib.import_module("video-summarization_amd.corpus")
    if wl == "corpus":
        # configs[3]: 50 + 25 ragged videos (tools/eval_corpus.py: lengths U[150,650] / U[100,650]),
        # the SAME corpus on every rank; a rank's shard is resident on its GPU, scores return to every rank
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        feats, targets, users = importlib.import_module("eval_corpus").corpus(seed=7)
        lengths = [int(f.shape[0]) for f in feats]
        mine = corpus_mod.plan_shards(lengths, world)[rank]
        videos = [f.to(dev) if i in set(mine) else f for i, f in enumerate(feats)]
        frames_per_step = sum(lengths)                       # whole job, fixed: strong scaling
        can_pack = d // H in (32, 64)
        gathered = None

        def step():
            corpus_mod.score_corpus(lambda xx, mm: model.score(xx, mm), videos, rank=rank, world=world, device=dev,
                                    max_frames=int(os.environ.get("VS_BENCH_CORPUS_FRAMES", "65536")), packed_fn=(lambda xx, ll: model.score_packed(xx, ll)) if can_pack else None,
                                    force_collective=dist is not None)
            return None
    else:
        g = torch.Generator(device="cpu").manual_seed(1234 + rank)       # disjoint videos per rank
        x_host = torch.randn(B, T, Din, generator=g).pin_memory()
        x = x_host.to(dev)
        frames_per_step = B * T * world                      # whole job, grows with the ranks: weak scaling
        gathered = torch.empty((world * B, T), dtype=torch.float32, device=dev) if dist is not None else None   # rank-major rows

        def step():
            logits, _hidden = model(x)
            if dist is not None:
                return dist.all_gather_into_tensor(gathered, logits.view(B, T), async_op=True)
            return None

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(k):
        fence()
        t0 = time.perf_counter()
        works = [step() for _ in range(k)]
        for w in works:
            if w is not None:
                w.wait()
        fence()
        return time.perf_counter() - t0

    with torch.no_grad():
        for _ in range(args.warmup):
            w = step()
            if w is not None:
                w.wait()
        dt = timed(args.steps)                    # the headline: profiling OFF

        # the same K steps again with the library's per-launch HIP events on (stage table / roofline only)
        lib.vs_profile_enable(1)
        timed(args.steps)
        stages = pkg._lib.profile_collect()
        lib.vs_profile_enable(0)

        # secondary: the same K steps with fp32 emulated on the f16 pipe (never `value`)
        emu = None
        if wl == "batch" and args.compute == "fp32" and not args.no_emulated and d // H in (32, 64, 128):
            exact_logits = model(x)[0].clone()
            model.set_compute_dtype("fp16x3")
            for _ in range(max(2, args.warmup // 2)):
                w = step()
                if w is not None:
                    w.wait()
            dt_emu = timed(args.steps)
            diff = (model(x)[0] - exact_logits).abs().max().item()
            model.set_compute_dtype("fp32")
            emu = (dt_emu, diff)
        # secondary: the same K steps with every product on the bf16 matrix pipe (reduced precision: never `value`)
        low = None
        if wl == "batch" and args.compute == "fp32" and not args.no_emulated and d <= 256 and d // H in (32, 64):
            exact_logits = model(x)[0].clone()
            model.set_compute_dtype("bf16")
            for _ in range(max(2, args.warmup // 2)):
                w = step()
                if w is not None:
                    w.wait()
            dt_low = timed(args.steps)
            diff = (model(x)[0] - exact_logits).abs().max().item()
            model.set_compute_dtype("fp32")
            low = (dt_low, diff)

        pcie_fps = pcie_overlap_fps = lat_ms = lat_splitk_ms = None
        train_ms = None
        eval_ms = None
        if wl == "corpus":           # the consumer of the gathered scores: sharded keyshot evaluation + all_reduce of four sums
            # timed on its own (not "a whole val_step minus a scoring pass"): score once, then only the evaluation of this
            # rank's shard (one C call: vs_eval_corpus) and the all_reduce of the four sums
            harness = importlib.import_module("video-summarization_amd.harness")
            can_pack = model.d_model // model.num_heads in (32, 64)
            sc = corpus_mod.score_corpus(lambda xx, mm: model.score(xx, mm), videos, rank=rank, world=world, device=dev,
                                         packed_fn=(lambda xx, ll: model.score_packed(xx, ll)) if can_pack else None)
            torch.cuda.synchronize()
            order = sorted(corpus_mod.plan_shards([int(v.shape[0]) for v in videos], world)[rank])
            harness.evaluate_shard(sc, targets, users, order)       # warm (library threads, page faults)
            fence()
            t1 = time.perf_counter()
            sums = harness.evaluate_shard(sc, targets, users, order)
            if world > 1:
                tt = torch.tensor(sums, dtype=torch.float64, device=dev)
                dist.all_reduce(tt)
                torch.cuda.synchronize()
            eval_ms = (time.perf_counter() - t1) * 1e3
        if wl == "batch" and not args.no_extras:
            # PCIe-inclusive rates (never `value`): pinned host batches -> device -> forward.  Serial = copy then
            # kernels on one stream; overlapped = corpus.score_host_batches (copy stream + compute stream).
            fence()
            t1 = time.perf_counter()
            for _ in range(3):
                model(x_host.to(dev, non_blocking=True))
            torch.cuda.synchronize()
            pcie_fps = 3 * B * T / (time.perf_counter() - t1)
            corpus = importlib.import_module("video-summarization_amd.corpus")
            nb = 8
            host_batches = [(x_host, None)] * nb
            corpus.score_host_batches(lambda xx, mm: model(xx)[0], host_batches[:2], dev)      # warm-up
            t1 = time.perf_counter()
            corpus.score_host_batches(lambda xx, mm: model(xx)[0], host_batches, dev)
            pcie_overlap_fps = nb * B * T / (time.perf_counter() - t1)
            # configs[1]: one T=320 video
            x1 = x[:1, :320].contiguous()
            for _ in range(10):
                model(x1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(200):
                model(x1)
            torch.cuda.synchronize()
            lat_ms = (time.perf_counter() - t1) / 200 * 1e3
            # ... and in the opt-in latency mode (SimNet.set_latency_mode: split-K Linears, keys split over a block's waves)
            lat_splitk_ms = None
            if hasattr(model, "set_latency_mode"):
                model.set_latency_mode(True)
                for _ in range(10):
                    model(x1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(200):
                    model(x1)
                torch.cuda.synchronize()
                lat_splitk_ms = (time.perf_counter() - t1) / 200 * 1e3
                model.set_latency_mode(False)
            # SURVEY §8(f) row 2, context only: one training step (HIP forward with dropout + masked-MSE loss + HIP backward) on
            # the same batch, exact fp32 and under set_train_dtype("bf16" / "fp16") (the counterparts of the reference's autocast)
            try:
                tmodel = pkg.SimNet(num_heads=H, d_model=d, num_layers=L, sparsity=0.0, dropout=0.3)
                tmodel.load_state_dict(sd)
                tmodel = tmodel.to(dev).train()
                ttarget = torch.rand(B, T, device=dev)
                tmask = torch.zeros(B, T, dtype=torch.bool, device=dev)

                def train_step():
                    with torch.enable_grad():          # (this block sits inside the scoring legs' no_grad)
                        pred, _h = tmodel(x, None)
                        loss_ = pkg.mse_with_mask_loss(pred, ttarget, tmask)
                        tmodel.zero_grad(set_to_none=True)
                        loss_.backward()

                train_ms = {}
                for mode in ("fp32", "bf16", "fp16"):
                    tmodel.set_train_dtype(mode)
                    for _ in range(3):
                        train_step()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(10):
                        train_step()
                    torch.cuda.synchronize()
                    train_ms[mode] = (time.perf_counter() - t1) / 10 * 1e3
                del tmodel
            except Exception as exc:       # an extra must never cost the headline line
                train_ms = {"error": repr(exc)}

    if dist is not None:
        t = torch.tensor([dt, emu[0] if emu else 0.0, emu[1] if emu else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t[0].item()
        if emu:
            emu = (t[1].item(), t[2].item())
    frames = frames_per_step * args.steps
    value = frames / dt

    out = None
    if rank == 0:
        if wl == "corpus":
            # ragged shard: per-stage FLOPs of ONE pass over rank 0's videos, divided by the launches that pass made
            # (one per packed batch and layer) = the average launch the event durations are averaged over
            mine_t = [lengths[i] for i in mine]
            Msh, Q = float(sum(mine_t)), float(sum(t * t for t in mine_t))
            per_pass = {"embed_pe": 2.0 * Msh * Din * d, "qkv_proj": L * 2.0 * Msh * d * 3 * d, "attention": L * 4.0 * Q * d,
                        "outproj_ln": L * 2.0 * Msh * d * d, "fc1_relu": L * 2.0 * Msh * d * 4 * d,
                        "fc2_ln_score": L * 2.0 * Msh * 4 * d * d + 2.0 * Msh * d}
            fl = {k: per_pass[k] / max(stages.get(k, (0, 0))[1] / args.steps, 1e-9) if stages.get(k, (0, 0))[1] else 0.0
                  for k in per_pass}
        else:
            fl = stage_flops(B, T, Din, d, H, L)
        # the bf16 mode folds stages into fewer kernels: a stage that was never launched did its
        # work inside another - the layer tail (reported as fc2_ln_score) carries the out-projection, fc1 and all but the
        # first QKV, the embedding kernel the first QKV; FLOPs per launch follow the work
        if stages.get("fc1_relu", (0, 0))[1] == 0 and stages.get("fc2_ln_score", (0, 0))[1]:
            fl["fc2_ln_score"] += fl["fc1_relu"]
            if stages.get("outproj_ln", (0, 0))[1] == 0:
                fl["fc2_ln_score"] += fl["outproj_ln"]
            nq = stages.get("qkv_proj", (0, 0))[1]
            if nq < stages["fc2_ln_score"][1]:
                per_step = stages["fc2_ln_score"][1] // L if L else 0      # timed + profiled steps recorded
                if nq == 0:
                    fl["embed_pe"] += fl["qkv_proj"]
                    fl["fc2_ln_score"] += fl["qkv_proj"] * (L - 1) / L
                elif per_step and nq == per_step:
                    fl["fc2_ln_score"] += fl["qkv_proj"] * (L - 1) / L
        table = {}
        for name, (ms, n) in stages.items():
            if n:
                avg = ms / n
                table[name] = {"launches": n, "avg_ms": round(avg, 4), "tflops": round(fl[name] / (avg * 1e-3) / 1e12, 2),
                               "share": 0.0}
        tot = sum(v["avg_ms"] * v["launches"] for v in table.values())
        for v in table.values():
            v["share"] = round(v["avg_ms"] * v["launches"] / tot, 3)
        dom = max(table, key=lambda k: table[k]["share"])
        flops_per_frame = 2 * Din * d + L * (24 * d * d + 4 * T * d) + 2 * d
        if wl == "corpus":            # corpus average: sum over videos of T (2 Din d + L (24 d^2 + 4 T d) + 2 d) / sum T
            flops_per_frame = sum(t * (2 * Din * d + L * (24 * d * d + 4 * t * d) + 2 * d) for t in lengths) / float(sum(lengths))
        # peak of the arithmetic the timed path used: fp32 MFMA, or f16 MFMA / 3 products (fp16x3), or bf16 MFMA
        peak = {"fp32": PEAK_F32_MFMA_TFLOPS, "fp16x3": PEAK_F16_MFMA_TFLOPS / 3, "bf16": PEAK_F16_MFMA_TFLOPS}[args.compute]
        traffic, traffic_src = (None, "not the measured configuration")
        if (wl, B, T, args.model, args.compute) == ("batch", 64, 1024, "A", "fp32"):
            traffic, traffic_src = measured_traffic(dom)
        roofline = {"bound": "mfma", "kernel": dom, "achieved": table[dom]["tflops"], "peak": round(peak, 1),
                    "unit": "TFLOP/s", "frac": round(table[dom]["tflops"] / peak, 4),
                    "traffic": traffic, "traffic_source": traffic_src, "hbm_gbps": None,
                    "flop_per_launch": fl[dom],
                    "whole_forward": {"flop_per_frame": flops_per_frame,
                                      "achieved": round(value / world * flops_per_frame / 1e12, 2),
                                      "frac": round(value / world * flops_per_frame / 1e12 / peak, 4)},
                    "stages": table}
        if traffic:
            # HBM GB/s of the dominant kernel: measured bytes per launch (PMC passes) / its live average duration
            roofline["hbm_gbps"] = round(traffic / (table[dom]["avg_ms"] * 1e-3) / 1e9, 1)
        cpu = None
        latency = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle.simnet_oracle import time_cpu_baseline, usable_cpus      # the checker, timed as the CPU "port"
            # `cores` = the host cores this job may use on the box (affinity / cgroup share); `threads` = the torch
            # thread count the sweep found fastest for the oracle (never more than `cores`)
            box_cores = usable_cpus()
            if wl == "long":         # one T=8192 video, one pass: four materialised [4, 8192, 8192] fp32 score tensors
                fps, threads, sample = time_cpu_baseline(sd, H, 1, T, 0.0, iters=1)
            elif wl == "corpus":     # three videos of the corpus' median length
                tm = sorted(lengths)[len(lengths) // 2]
                fps, threads, sample = time_cpu_baseline(sd, H, 3, tm, args.cpu_seconds)
            else:
                fps, threads, sample = time_cpu_baseline(sd, H, 8, T, args.cpu_seconds)
            cpu = {"value": round(fps, 1), "unit": "frames/s", "cores": box_cores, "threads": threads, "kind": "port",
                   "sample": "oracle/simnet_oracle.py (materialised [B,H,T,T] softmax, torch CPU fp32), " + sample}
            if wl == "batch":
                fps1, threads1, sample1 = time_cpu_baseline(sd, H, 1, 320, 0.0, iters=20)
                cpu["single_video"] = {"value": round(fps1, 1), "unit": "frames/s", "ms_per_video": round(320 / fps1 * 1e3, 3),
                                       "cores": box_cores, "threads": threads1, "sample": sample1}
        if lat_ms is not None:
            latency = {"workload": "configs[1]: one video, T=320, D=1024, M-%s" % args.model, "gpu_ms": round(lat_ms, 4),
                       "gpu_frames_per_s": round(320 / lat_ms * 1e3, 1),
                       "cpu_ms": cpu["single_video"]["ms_per_video"] if cpu else None}
            if lat_splitk_ms is not None:       # opt-in latency mode: deterministic, 1e-4 of the goldens, not the default kernels' bits
                latency["latency_mode_gpu_ms"] = round(lat_splitk_ms, 4)
                latency["latency_mode"] = "SimNet.set_latency_mode(): VS_FLAG_SPLITK (split-K embedding / out-projection / fc2 + row LayerNorm, keys split over a block's waves)"
        par = "%s all_gather of scores" % ("RCCL" if backend == "nccl" else (backend or "no"))
        if wl == "corpus":
            workload = ("configs[3]: TVSum+SumMe-shaped corpus, %d ragged videos / %d frames (T 100..650, D=1024) dealt to %d "
                        "GPU(s) by cost, packed batches, scorer cfg M-%s (heads %d, d_model %d, layers %d), sigmoid scores "
                        "returned to every rank" % (len(lengths), sum(lengths), world, args.model, H, d, L))
            cfg = {"workload": workload, "videos": len(lengths), "frames_per_step": frames_per_step,
                   "parallelism": "videos sharded over %d GPU(s) (strong scaling), %s" % (world, par)}
            metric = "frames/sec scored (whole node), synthetic TVSum+SumMe-shaped ragged corpus [75 videos, D=1024]"
        else:
            workload = ("configs[%d]: B=%d videos x T=%d frames x D=%d per GPU, scorer cfg M-%s (heads %d, d_model %d, layers %d), "
                        "logits + hidden state" % (4 if wl == "long" else 2, B, T, Din, args.model, H, d, L))
            cfg = {"workload": workload, "global_batch": B * world, "frames_per_step": frames_per_step,
                   "parallelism": "videos sharded over %d GPU(s), %s" % (world, par)}
            metric = "frames/sec scored (whole node), synthetic [B,T=%d,D=%d]" % (T, Din)
        out = {
            "metric": metric,
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if wl == "corpus" else "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "fp16x3": "f32 emulated as 3 x f16 MFMA (hi+lo operand split), f32 accumulate",
                      "bf16": "bf16 MFMA operands, f32 accumulate"}[args.compute],
            "data": "synthetic",
            "config": cfg,
            "collective_backend": backend, "ranks_seen": ranks_seen,
            "pcie_inclusive_value": round(pcie_fps * world, 1) if pcie_fps else None,
            "pcie_inclusive_overlapped_value": round(pcie_overlap_fps * world, 1) if pcie_overlap_fps else None,
            "roofline": roofline, "cpu_baseline": cpu, "latency": latency,
        }
        if eval_ms is not None:
            out["eval_ms"] = round(eval_ms, 3)
        if train_ms:
            out["training_step"] = ({"workload": "forward under autograd (dropout 0.3) + masked MSE + backward on the same B x T batch, HIP "
                                                 "kernels (include/vs_train.h); fp32 = exact, bf16 / fp16 = SimNet.set_train_dtype(...)",
                                     "fp32_ms": round(train_ms["fp32"], 3), "bf16_ms": round(train_ms["bf16"], 3),
                                     "fp16_ms": round(train_ms["fp16"], 3),
                                     "fp32_frames_per_s": round(B * T / train_ms["fp32"] * 1e3, 1),
                                     "bf16_frames_per_s": round(B * T / train_ms["bf16"] * 1e3, 1),
                                     "fp16_frames_per_s": round(B * T / train_ms["fp16"] * 1e3, 1)}
                                    if "error" not in train_ms else train_ms)
        if emu:
            ev = frames / emu[0]
            out["emulated_f32"] = {
                "mode": "fp16x3: every product as 3 f16 MFMAs over hi+lo operand halves, f32 accumulate (opt-in "
                        "SimNet.set_compute_dtype('fp16x3'); parity tests hold it to the same 1e-4 goldens)",
                "value": round(ev, 1), "unit": "frames/s", "ms_per_step": round(emu[0] / args.steps * 1e3, 4),
                "speedup_vs_value": round(ev / value, 3), "max_abs_logit_diff_vs_exact": emu[1],
                "f32_equivalent_tflops": round(ev / world * flops_per_frame / 1e12, 2)}
        if low:
            lv = frames / low[0]
            out["bf16_mode"] = {
                "mode": "bf16: every product on the bf16 matrix pipe, fp32 accumulation / residual / LayerNorm / softmax "
                        "statistics (opt-in SimNet.set_compute_dtype('bf16'); REDUCED precision: logits within ~4e-3, "
                        "outside the 1e-4 parity bar - context, not the headline)",
                "value": round(lv, 1), "unit": "frames/s", "ms_per_step": round(low[0] / args.steps * 1e3, 4),
                "speedup_vs_value": round(lv / value, 3), "max_abs_logit_diff_vs_exact": low[1]}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
