# Round checkpoint on the GPU box: the whole -m gpu suite, smoke(), the headline bench, the HBM-traffic PMC passes, the
# rocprofv3 kernel-trace of the same bench command, PMC summaries, M-B and training profiles.  Outputs: gpurun_out/rNN/
# usage: gpu_round_check.sh [tag] [part]   part 1 = parity + headline measurements, part 2 = the other modes' profiles,
# no part = both (longer than one 20-minute gpurun call since round 3)
set -e
TAG=${1:-r04z}
PART=${2:-0}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
if [ "$PART" != "2" ]; then
# HBM-traffic passes first: the bench contract test below wants a traffic file measured on the sources that are loaded
bash tools/collect_traffic.sh > gpurun_out/$TAG/traffic.log 2>&1
cp gpurun_out/traffic.json gpurun_out/$TAG/traffic.json
cp gpurun_out/traffic.json profiles/r04_hbm_traffic.json      # (on the box; copy gpurun_out/$TAG/traffic.json home as well)
python -m pytest tests -x -q -m gpu > gpurun_out/$TAG/gputest.log 2>&1 || { tail -40 gpurun_out/$TAG/gputest.log; exit 1; }
tail -3 gpurun_out/$TAG/gputest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
set +e          # measurements below: one failing profile step must not skip the rest
python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
cat gpurun_out/$TAG/bench.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_bench.json 2>$GRAFT_REPO_ROOT/gpurun_out/$TAG/prof.err )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_mb -- python3 $GRAFT_REPO_ROOT/bench.py --model B --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_bench_mb.json 2>$GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_mb.err )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_train -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py --shapes 64x1024 --iters 5 > /dev/null 2>&1 )
python tools/bench_train.py --torch > gpurun_out/$TAG/bench_train.txt 2>&1
python tools/bench_train.py --model B --shapes 4x320,64x1024 >> gpurun_out/$TAG/bench_train.txt 2>&1
bash tools/pmc_cmd.sh gpurun_out/$TAG/pmc_summary.txt bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
bash tools/pmc_cmd.sh gpurun_out/$TAG/pmc_summary_mb.txt bench.py --model B --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-emulated > /dev/null 2>&1
python tools/bench_configs.py > gpurun_out/$TAG/bench_configs.txt 2>&1
fi
if [ "$PART" = "1" ]; then echo "part 1 done"; exit 0; fi
set +e
# the opt-in modes: configs[4] (8 x 8192 x 2048) and the headline shape in fp32 / bf16 / fp16x3, the layer-tail kernel's
# ablations, kernel trace + PMC summary of the bf16 mode at configs[4]
python tools/bench_long.py 8 8192 fp32,bf16,fp16x3 > gpurun_out/$TAG/bench_long.txt 2>&1
VS_BENCH_D=1024 python tools/bench_long.py 64 1024 fp32,bf16,fp16x3 > gpurun_out/$TAG/bench_headline_modes.txt 2>&1
VS_MLP_ABLS=1,2,4,7,8 python tools/bench_mlp_fused.py > gpurun_out/$TAG/mlp_fused_ablations.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_long_bf16 -- python3 $GRAFT_REPO_ROOT/tools/bench_long.py 8 8192 bf16 > /dev/null 2>&1 )
bash tools/pmc_cmd.sh gpurun_out/$TAG/pmc_summary_long_bf16.txt tools/bench_long.py 8 8192 bf16 > /dev/null 2>&1
# round 3: the other bench workloads, wide models, attention timelines / ablations / issue probe, raw PMC of the bf16 attention
python bench.py --workload corpus --steps 20 --warmup 3 > gpurun_out/$TAG/bench_corpus.json 2> gpurun_out/$TAG/bench_corpus.err
python bench.py --workload long --steps 20 --warmup 3 > gpurun_out/$TAG/bench_long.json 2> gpurun_out/$TAG/bench_long.err
python tools/bench_wide.py > gpurun_out/$TAG/bench_wide.txt 2>&1
python tools/bench_mb_modes.py > gpurun_out/$TAG/mb_modes.txt 2>&1
python tools/diag_attention.py > gpurun_out/$TAG/attention_timeline_exact.txt 2>&1
python tools/diag_attention.py abl > gpurun_out/$TAG/attn_bf16_ablations.txt 2>&1
./tools/valu_probe > gpurun_out/$TAG/valu_probe.txt 2>&1 || true
# round 3, second half: the bf16-operand GEMM, M-B / wide models in bf16 mode (kernel trace + PMC), the bf16 training step
python tools/bench_gemm16.py > gpurun_out/$TAG/gemm16.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_mb_bf16 -- python3 $GRAFT_REPO_ROOT/tools/prof_mb_bf16.py > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_mb_bf16.txt 2>&1 )
bash tools/pmc_cmd.sh gpurun_out/$TAG/pmc_summary_mb_bf16.txt tools/prof_mb_bf16.py > /dev/null 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_train_bf16 -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py --shapes 64x1024 --only-bf16 --iters 10 > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_train_bf16.txt 2>&1 )
bash tools/pmc_cmd.sh gpurun_out/$TAG/pmc_summary_train_bf16.txt tools/bench_train.py --shapes 64x1024 --only-bf16 --iters 4 > /dev/null 2>&1
# round 4: reference-sized calls (default kernels, then the opt-in latency mode), the one-wave-per-SIMD attention check + timings
{ python tools/latency_breakdown.py 1 320; python tools/latency_breakdown.py 1 320 splitk; python tools/latency_breakdown.py 4 320; python tools/latency_breakdown.py 4 320 splitk; } > gpurun_out/$TAG/latency_scoring.txt 2>&1
python tools/check_attn_w64.py > gpurun_out/$TAG/check_attn_w64.txt 2>&1
python tools/train_step_breakdown.py 4 320 --profile > gpurun_out/$TAG/latency_train_step.txt 2>&1
{ python tools/latency_graph.py; VS_LAT_MODEL=B python tools/latency_graph.py; VS_LAT_MODEL=C python tools/latency_graph.py; } > gpurun_out/$TAG/latency_graph.txt 2>&1
echo done
