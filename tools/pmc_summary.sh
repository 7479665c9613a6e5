#!/bin/bash
# Per-kernel MFMA-pipe utilisation (cycles), effective clock and instruction mix per MFMA for one bench run.
# GPU box:  bash tools/pmc_summary.sh > gpurun_out/pmc_summary.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/pmc_all
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_MFMA --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_all -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/pmc_all.log 2>&1
python3 - <<PY
import csv, collections, glob
f=glob.glob('$ROOT/gpurun_out/pmc_all/*/*counter_collection.csv')[0]
per=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    per.setdefault(r['Dispatch_Id'],{'name':r['Kernel_Name'].replace('void (anonymous namespace)::','').split('(')[0],'t0':int(r['Start_Timestamp']),'t1':int(r['End_Timestamp'])})[r['Counter_Name']]=float(r['Counter_Value'])
agg=collections.defaultdict(list)
for v in per.values():
    if v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)>0:
        agg[(v['name'], round((v['t1']-v['t0'])/1e5))].append(v)
print("bench.py B=64 T=1024 M-A under rocprofv3 --pmc (profiled passes run ~3-5 % slower); util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); VALU = non-MFMA vector instructions per MFMA (SQ_INSTS_VALU - SQ_INSTS_MFMA); cyc/MFMA = busy cycles per MFMA instruction")
for (n,_),l in sorted(agg.items()):
    v=l[len(l)//2]; dur=v['t1']-v['t0']; cyc=v['GRBM_GUI_ACTIVE']/8; nm=v.get('SQ_INSTS_MFMA',0) or v['SQ_VALU_MFMA_BUSY_CYCLES']/64   # MFMA instructions (SQ_INSTS_MFMA; rounds 1-2 divided busy cycles by 64, which doubled every per-MFMA figure of the 32-cycle bf16 / f16 instructions)
    print("%-30s n=%3d dur %4.0f us clk %.2f GHz mfma-util %.3f | per MFMA: VALU %.2f SALU %.2f LDS %.2f VMEM_RD %.3f VMEM_WR %.3f" % (n,len(l),dur/1e3,cyc/dur,v['SQ_VALU_MFMA_BUSY_CYCLES']/(cyc*1024),(v['SQ_INSTS_VALU']-nm)/nm,v['SQ_INSTS_SALU']/nm,v['SQ_INSTS_LDS']/nm,v['SQ_INSTS_VMEM_RD']/nm,v['SQ_INSTS_VMEM_WR']/nm))
PY
