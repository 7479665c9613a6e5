set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
python -m pytest tests -x -q -m gpu > gpurun_out/r02a/gputest.log 2>&1 || { tail -40 gpurun_out/r02a/gputest.log; exit 1; }
tail -3 gpurun_out/r02a/gputest.log
python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
cat gpurun_out/r02a/bench.json
bash tools/collect_traffic.sh > gpurun_out/r02a/traffic.log 2>&1
cp gpurun_out/traffic.json gpurun_out/r02a/traffic.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02a/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/r02a/prof_bench.json 2>$GRAFT_REPO_ROOT/gpurun_out/r02a/prof.err
cd $GRAFT_REPO_ROOT && bash tools/pmc_summary.sh > gpurun_out/r02a/pmc_summary.txt 2>&1
echo done
