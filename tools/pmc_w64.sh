#!/bin/bash
# Counters of the one-wave-per-SIMD attention and of its timing ablations (diagnostic library), one rocprofv3 pass each.
#   bash tools/pmc_w64.sh <out.txt> "0 2 4 8 1" "SQ_WAVE_CYCLES SQ_WAIT_ANY ..."
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$1; ABLS=$2; CTRS=$3
cd /tmp && export TMPDIR=/tmp
for a in $ABLS; do
  rm -rf $ROOT/gpurun_out/pmc_w64
  timeout -k 10 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_w64 -- python3 $ROOT/tools/check_attn_w64.py ablone $a > $ROOT/gpurun_out/pmc_w64.log 2>&1
  python3 - <<PY >> $ROOT/$OUT
import csv, collections, glob
f=glob.glob('$ROOT/gpurun_out/pmc_w64/*/*counter_collection.csv')[0]
per=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    per.setdefault(r['Dispatch_Id'],{'name':r['Kernel_Name'],'t0':int(r['Start_Timestamp']),'t1':int(r['End_Timestamp'])})[r['Counter_Name']]=float(r['Counter_Value'])
l=[v for v in per.values() if 'attn_fwd_bf16_w64' in v['name']]
l=sorted(l,key=lambda v:v['t1']-v['t0']); v=l[len(l)//2]
ctrs="$CTRS".split(); base=v.get('SQ_WAVE_CYCLES')
iters=1024*4*128.0
print("abl %4s dur %6.1f us | " % ("$a",(v['t1']-v['t0'])/1e3) + " ".join("%s=%.0f/it%s" % (c.replace('SQ_',''), 4*v.get(c,0)/iters if c not in ('SQ_INSTS_VALU','SQ_INSTS_MFMA','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_VALU_MFMA_BUSY_CYCLES') else v.get(c,0)/iters, (" (%.3f)" % (v.get(c,0)/base) if base and c!='SQ_WAVE_CYCLES' else "")) for c in ctrs))
PY
done
cat $ROOT/$OUT
