#!/bin/bash
# Per-kernel MFMA-pipe utilisation, effective clock and instruction mix per MFMA for an arbitrary python command.
# GPU box:  bash tools/pmc_cmd.sh <out.txt> tools/bench_train.py --shapes 64x1024 --iters 3
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/pmc_cmd
timeout -k 10 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_MFMA --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_cmd -- python3 $ROOT/"$@" > $ROOT/gpurun_out/pmc_cmd.log 2>&1
python3 - <<PY > $ROOT/$OUT
import csv, collections, glob
f=glob.glob('$ROOT/gpurun_out/pmc_cmd/*/*counter_collection.csv')[0]
per=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    per.setdefault(r['Dispatch_Id'],{'name':r['Kernel_Name'].replace('void (anonymous namespace)::','').replace('(anonymous namespace)::','').split('(')[0],'t0':int(r['Start_Timestamp']),'t1':int(r['End_Timestamp'])})[r['Counter_Name']]=float(r['Counter_Value'])
agg=collections.defaultdict(list)
for v in per.values():
    if v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)>0:
        agg[v['name']].append(v)
print("command: $@  (rocprofv3 --pmc; profiled passes run a few % slower); util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); VALU = non-MFMA vector instructions per MFMA (SQ_INSTS_VALU - SQ_INSTS_MFMA); cyc/MFMA = busy cycles per MFMA instruction; longest-duration half of each kernel's dispatches")
for n,l in sorted(agg.items()):
    l=sorted(l,key=lambda v:v['t1']-v['t0'])[len(l)//2:]
    v=l[len(l)//2]; dur=v['t1']-v['t0']; cyc=v['GRBM_GUI_ACTIVE']/8; nm=v.get('SQ_INSTS_MFMA',0) or v['SQ_VALU_MFMA_BUSY_CYCLES']/64   # MFMA instructions (SQ_INSTS_MFMA; rounds 1-2 divided busy cycles by 64, which doubled every per-MFMA figure of the 32-cycle bf16 / f16 instructions)
    print("%-34s n=%3d dur %5.0f us clk %.2f GHz mfma-util %.3f | per MFMA: VALU %.2f SALU %.2f LDS %.2f VMEM_RD %.3f VMEM_WR %.3f WAIT_ANY/cyc %.2f" % (n,len(l),dur/1e3,cyc/dur,v['SQ_VALU_MFMA_BUSY_CYCLES']/(cyc*1024),(v['SQ_INSTS_VALU']-nm)/nm,v['SQ_INSTS_SALU']/nm,v['SQ_INSTS_LDS']/nm,v['SQ_INSTS_VMEM_RD']/nm,v['SQ_INSTS_VMEM_WR']/nm, v['SQ_WAIT_ANY']/ (cyc*1024) ))
PY
cat $ROOT/$OUT
