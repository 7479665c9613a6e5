#!/bin/bash
# HBM bytes per forward of the bf16 mode at configs[4] (8 x 8192 x 2048), per kernel, from the TCC counters
# (FETCH_SIZE / WRITE_SIZE in separate passes, KiB, FETCH x2 on gfx950: MI355X_MICROARCH.md, HBM), for the current
# kernels and with the round-2 steps switched back (VS_LP_STORE32=1 VS_LP_MLP_UNFUSED=1 = the round-1 data path).
# GPU box:  bash tools/traffic_long.sh gpurun_out/long_traffic.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/$1
cd /tmp && export TMPDIR=/tmp
: > $OUT
for cfg in current round1; do
  if [ $cfg = round1 ]; then export VS_LP_STORE32=1 VS_LP_MLP_UNFUSED=1; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$ROOT/gpurun_out/tl_${cfg}_$c
    rm -rf $d
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $ROOT/tools/bench_long.py 8 8192 bf16 > $ROOT/gpurun_out/tl_${cfg}_$c.log 2>&1
  done
  python3 - <<PY >> $OUT
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$ROOT/gpurun_out/tl_${cfg}_%s/*/*counter_collection.csv" % c)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
            acc[n][c].append(float(r["Counter_Value"]))
print("== bf16 mode, B=8 T=8192 D=2048, $cfg kernels: HBM MB per launch (2 x FETCH_SIZE + WRITE_SIZE), launches per forward, MB per forward")
tot = 0.0
fw = 23     # forwards the run makes (3 warm-up + 10 timed + 10 profiled)
for n, v in sorted(acc.items()):
    if len(v["FETCH_SIZE"]) < fw or n.startswith("pack_") or "distribution" in n:
        continue
    f = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]); w = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
    mb = (2 * f + w) * 1024 / 1e6
    per = len(v["FETCH_SIZE"]) / fw
    tot += mb * per
    print("  %-44s %8.1f MB x %4.1f = %8.1f MB" % (n, mb, per, mb * per))
print("  total per forward: %.2f GB" % (tot / 1e3))
PY
done
cat $OUT
