#!/bin/bash
# Raw PMC counters per kernel (median dispatch of the longest-duration half) for an arbitrary python command.
#   bash tools/pmc_raw.sh <out.txt> "<counter list>" tools/bench_long.py 8 8192 bf16
# Ratios to SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE are printed where both are in the list.  One pass = at most 8 SQ counters.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$1; shift
CTRS=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/pmc_raw
timeout -k 10 600 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_raw -- python3 $ROOT/"$@" > $ROOT/gpurun_out/pmc_raw.log 2>&1
python3 - <<PY >> $ROOT/$OUT
import csv, collections, glob
f=glob.glob('$ROOT/gpurun_out/pmc_raw/*/*counter_collection.csv')[0]
per=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    per.setdefault(r['Dispatch_Id'],{'name':r['Kernel_Name'].replace('void (anonymous namespace)::','').replace('(anonymous namespace)::','').split('(')[0],'t0':int(r['Start_Timestamp']),'t1':int(r['End_Timestamp'])})[r['Counter_Name']]=float(r['Counter_Value'])
agg=collections.defaultdict(list)
for v in per.values(): agg[v['name']].append(v)
ctrs="$CTRS".split()
print("command: $@ | counters: $CTRS")
for n,l in sorted(agg.items()):
    l=sorted(l,key=lambda v:v['t1']-v['t0'])[len(l)//2:]
    v=l[len(l)//2]; dur=(v['t1']-v['t0'])/1e3
    if dur < 20: continue
    base=v.get('SQ_WAVE_CYCLES') or v.get('SQ_BUSY_CU_CYCLES')
    s=" ".join("%s=%.4g%s" % (c, v.get(c,0), (" (%.3f)" % (v.get(c,0)/base) if base and c!='SQ_WAVE_CYCLES' else "")) for c in ctrs)
    print("%-44s n=%3d dur %7.1f us | %s" % (n[:44], len(l), dur, s))
PY
tail -20 $ROOT/$OUT
