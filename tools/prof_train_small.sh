cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/train_small -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py --shapes 4x320 --iters 50 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/train_small/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:26]:
    print("%-64s calls %5s avg %8.1f us  %5.1f%%" % (r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::","")[:64], r["Calls"], float(r["AverageNs"]) / 1e3, 100*float(r["TotalDurationNs"])/tot))
PY
