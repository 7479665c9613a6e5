#!/bin/bash
# MFMA-pipe utilisation (cycles) of the attention kernel for a list of ablation masks (GPU box).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for a in "$@"; do
  if [ "$a" = 0 ]; then unset VS_ATTN_ABL; else export VS_ATTN_ABL=$a; fi
  rm -rf $ROOT/gpurun_out/pmc_abl$a
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_abl$a -- python3 $ROOT/tools/bench_stage.py attention 30 > $ROOT/gpurun_out/pmc_abl$a.log 2>&1
done
python3 - "$@" <<PY
import csv, collections, glob, sys
for a in sys.argv[1:]:
    f=glob.glob('$ROOT/gpurun_out/pmc_abl%s/*/*counter_collection.csv'%a)[0]
    per=collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        per.setdefault(r['Dispatch_Id'],{'name':r['Kernel_Name'],'t0':int(r['Start_Timestamp']),'t1':int(r['End_Timestamp'])})[r['Counter_Name']]=float(r['Counter_Value'])
    l=[v for v in per.values() if 'attn' in v['name']]
    v=l[-1]; dur=v['t1']-v['t0']; cyc=v['GRBM_GUI_ACTIVE']/8
    print("ABL %4s: dur %.0f us clk %.2f GHz  mfma util %.3f | wait_any %.2f wait_inst %.2f active %.2f | VALU insts/wave-MFMA %.2f" % (a, dur/1e3, cyc/dur, v['SQ_VALU_MFMA_BUSY_CYCLES']/(cyc*1024), v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES'], v['SQ_WAIT_INST_ANY']/v['SQ_WAVE_CYCLES'], v['SQ_ACTIVE_INST_ANY']/v['SQ_WAVE_CYCLES'], v['SQ_INSTS_VALU']/(v['SQ_VALU_MFMA_BUSY_CYCLES']/64)))
PY
