#!/bin/bash
# HBM traffic per kernel launch from the TCC counters, as MI355X_MICROARCH.md §HBM prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass), values in KiB,
# FETCH_SIZE doubled on gfx950 for wide (16 B/lane) coalesced reads.  Run on the GPU box:
#   bash tools/collect_traffic.sh   ->  gpurun_out/traffic_{fetch,write}/..., gpurun_out/traffic.json
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=$ROOT/gpurun_out/traffic_$c
  rm -rf $d
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-emulated --no-extras > $ROOT/gpurun_out/traffic_$c.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections, sys
sys.path.insert(0, "$ROOT")
import bench
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$ROOT/gpurun_out/traffic_%s/*/*counter_collection.csv" % c)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            name = r["Kernel_Name"].replace("void (anonymous namespace)::", "")
            acc[name.split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k][c + "_KiB_avg"] = sum(v) / len(v)
        out[k]["launches"] = len(v)
for k, v in out.items():
    f, w = v.get("FETCH_SIZE_KiB_avg", 0.0), v.get("WRITE_SIZE_KiB_avg", 0.0)
    v["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0      # gfx950: FETCH_SIZE counts 64 B per 128-B request
# provenance: the kernel sources these bytes were measured on (bench.py emits `traffic` only when they match)
out["_meta"] = {"csrc_sha256_16": bench.kernel_source_hash(),
                "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-emulated --no-extras",
                "units": "FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM)"}
json.dump(out, open("$ROOT/gpurun_out/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
