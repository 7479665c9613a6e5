"""bench.py's own rank launcher (`python bench.py --gpus N` without torchrun): CPU dry run of the launch plumbing
(gloo, world size 2) and, on the GPU box, a two-rank rehearsal of the real control flow on one GPU."""
import json
import os
import subprocess
import sys

import pytest

import tolerances as tol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_plain_python_invocation_starts_its_own_ranks():
    """`python bench.py --gpus 2` (no torchrun env) must start 2 ranks itself and print ONE rank-0 line."""
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"VS_BENCH_DRYRUN": "1"}, 300)
    assert out == {"dryrun": True, "n_gpus": 2, "ranks": [0, 1], "self_launched": True}


def test_single_rank_does_not_relaunch():
    out = _run(["--gpus", "1"], {"VS_BENCH_DRYRUN": "1"}, 120)
    assert out["n_gpus"] == 1 and out["self_launched"] is False


@pytest.mark.gpu
def test_two_rank_rehearsal_through_the_self_launcher():
    """The real N > 1 control flow (per-rank batches, async all_gather per step, max-over-ranks timing), two ranks on the
    one GPU over gloo (VS_BENCH_REHEARSE=1: RCCL refuses two ranks on one device).  Numbers are meaningless here."""
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "2", "--frames", "256", "--no-cpu-baseline",
                "--no-emulated", "--no-extras"], {"VS_BENCH_REHEARSE": "1"}, 600)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["value"] > 0
    assert out["collective_backend"] == "gloo" and out["roofline"]["stages"]["attention"]["launches"] == 3 * 4


@pytest.mark.gpu
def test_bench_runs_its_rccl_gather_on_one_gpu():
    """VS_BENCH_FORCE_DIST=1: one rank, backend 'nccl' - every step's async all_gather_into_tensor of the [B,T] scores
    and the barriers of the timed region run over RCCL itself (the transport the 8-GPU run uses)."""
    out = _run(["--gpus", "1", "--steps", "5", "--warmup", "2", "--batch", "4", "--frames", "256", "--no-cpu-baseline",
                "--no-emulated", "--no-extras"], {"VS_BENCH_FORCE_DIST": "1"}, 600)
    assert out["n_gpus"] == 1 and out["collective_backend"] == "nccl" and out["value"] > 0
    assert "RCCL all_gather" in out["config"]["parallelism"]


@pytest.mark.gpu
def test_bench_line_carries_the_contract_fields_and_a_traceable_traffic_figure():
    """One short run at the bench's own shape: every field the driver reads is there, `roofline.traffic` comes from the
    committed PMC file (its source hash must match the loaded kernels - a stale file reads as null and fails here, so a
    kernel edit without `tools/collect_traffic.sh` is caught), and the secondary legs (fp16x3, bf16) are printed beside,
    never as, `value`."""
    out = _run(["--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-extras"], {}, 900)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["dtype"] == "f32" and out["n_gpus"] == 1 and out["steps"] == 5 and "workload" in out["config"]
    rf = out["roofline"]
    assert rf["bound"] == "mfma" and 0.5 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["traffic"] is not None and rf["traffic"] > 2.0e8, rf["traffic_source"]      # attention: ~268 MB per launch
    assert out["emulated_f32"]["value"] > out["value"] and out["emulated_f32"]["max_abs_logit_diff_vs_exact"] < 1e-4
    assert out["bf16_mode"]["value"] > out["emulated_f32"]["value"] and out["bf16_mode"]["max_abs_logit_diff_vs_exact"] < tol.BF16_LOGIT_TOL


@pytest.mark.gpu
def test_bench_extras_do_not_fail_silently():
    """The extras of the bench line (latency of one T=320 video - default kernels and latency mode -, one training step in
    fp32 / bf16 / fp16) are wrapped so that they can never cost the headline line; a genuine failure would then only show
    as an {"error": ...} field.  This run keeps the extras on and asserts on them."""
    out = _run(["--steps", "3", "--warmup", "1", "--batch", "8", "--frames", "512", "--no-cpu-baseline", "--no-emulated"], {}, 900)
    ts = out.get("training_step")
    assert ts is not None and "error" not in ts, ts
    assert ts["fp32_ms"] > 0 and ts["bf16_ms"] > 0 and ts["fp16_ms"] > 0
    lat = out.get("latency")
    assert lat is not None and 0 < lat["latency_mode_gpu_ms"] < lat["gpu_ms"], lat


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["corpus", "long"])
def test_two_rank_rehearsal_of_the_other_workloads(workload):
    """configs[3] (`--workload corpus`: strong scaling of the 75-video corpus, scores gathered to every rank, sharded
    evaluation timed beside) and configs[4] (`--workload long`: bf16) under the same self-launcher and JSON contract;
    two ranks on the one GPU over gloo (numbers meaningless), plus the one-rank form."""
    extra = ["--workload", workload, "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    if workload == "long":
        extra += ["--batch", "1", "--frames", "512"]
    two = _run(["--gpus", "2"] + extra, {"VS_BENCH_REHEARSE": "1"}, 900)
    one = _run(["--gpus", "1"] + extra, {}, 900)
    for out, n in ((two, 2), (one, 1)):
        assert out["n_gpus"] == n and out["value"] > 0 and out["ranks_seen"] == list(range(n))
        assert out["roofline"]["frac"] > 0 and out["roofline"]["stages"]["attention"]["launches"] > 0
        if workload == "corpus":
            assert out["scaling"] == "strong" and out["config"]["videos"] == 75 and out["config"]["frames_per_step"] == 30568
            assert out["eval_ms"] is not None and "configs[3]" in out["config"]["workload"]
        else:
            assert out["scaling"] == "weak" and out["dtype"].startswith("bf16") and "configs[4]" in out["config"]["workload"]
            assert out["config"]["frames_per_step"] == 512 * n and out["roofline"]["peak"] == 2500.0
