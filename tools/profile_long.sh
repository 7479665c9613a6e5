#!/bin/bash
# BASELINE configs[4] (B=8, T=8192, D=2048) under rocprofv3: kernel-trace stats, then one PMC pass with the
# MFMA-pipe busy cycles per kernel.  GPU box:  bash tools/profile_long.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/long_stats $ROOT/gpurun_out/long_pmc
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/long_stats -- python3 $ROOT/tools/bench_long.py > $ROOT/gpurun_out/long_stats.log 2>&1 || exit 1
cp $(ls -t $ROOT/gpurun_out/long_stats/*/*kernel_stats.csv | head -1) $ROOT/gpurun_out/long_kernel_stats.csv
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $ROOT/gpurun_out/long_pmc -- python3 $ROOT/tools/bench_long.py > $ROOT/gpurun_out/long_pmc.log 2>&1 || exit 1
python3 - <<PY
import csv, collections, glob
f=sorted(glob.glob('$ROOT/gpurun_out/long_pmc/*/*counter_collection.csv'))[-1]
per=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    per.setdefault(r['Dispatch_Id'],{'name':r['Kernel_Name'].replace('void (anonymous namespace)::','').split('(')[0],'t0':int(r['Start_Timestamp']),'t1':int(r['End_Timestamp'])})[r['Counter_Name']]=float(r['Counter_Value'])
agg=collections.defaultdict(list)
for v in per.values():
    if v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)>0:
        agg[(v['name'], round((v['t1']-v['t0'])/2e5))].append(v)
print("tools/bench_long.py (B=8 T=8192 D=2048 M-A; fp32, bf16-attention, bf16 modes) under rocprofv3 --pmc; util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); VALU count includes the MFMAs")
for (n,_),l in sorted(agg.items()):
    v=l[len(l)//2]; dur=v['t1']-v['t0']; cyc=v['GRBM_GUI_ACTIVE']/8
    print("%-44s n=%3d dur %6.0f us clk %.2f GHz mfma-busy %.3f | VALU insts/wave-cycle %.3f  wait_any %.2f" % (n,len(l),dur/1e3,cyc/dur,v['SQ_VALU_MFMA_BUSY_CYCLES']/(cyc*1024),v['SQ_INSTS_VALU']/max(v['SQ_WAVE_CYCLES'],1)*4, v['SQ_WAIT_ANY']/max(v['SQ_WAVE_CYCLES'],1)))
PY
