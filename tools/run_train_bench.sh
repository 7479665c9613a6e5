set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_train.py -q -k wgrad 2>&1 | tail -2
python tools/bench_train.py --torch > gpurun_out/bench_train.log 2>&1
cat gpurun_out/bench_train.log
python tools/bench_train.py --dropout 0 --shapes 64x1024 >> gpurun_out/bench_train.log 2>&1
tail -1 gpurun_out/bench_train.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/train_prof -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py --shapes 64x1024 --iters 5 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/train_prof/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:28]:
    print("%-70s calls %5s avg %9.1f us  %5s%%" % (r["Name"].replace("void (anonymous namespace)::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
