#!/bin/bash
# PMC passes over one ablation configuration: bash scripts/pmc_abl.sh <dbg>   (GPU box, repo root)
set -e
DBG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_abl_$DBG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -o pmc -- python3 $ROOT/scripts/abl_one.py $DBG > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob
from collections import defaultdict
c=defaultdict(lambda:[0,0.0]); dur=[]
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3_ws_kernel" in r["Kernel_Name"] or "conv_fused_kernel" in r["Kernel_Name"]:
            k=c[r["Counter_Name"]]; k[0]+=1; k[1]+=float(r["Counter_Value"])
for f in glob.glob("$OUT/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3_ws_kernel" in r["Kernel_Name"] or "conv_fused_kernel" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("dbg $DBG avg_us", sum(dur)/max(len(dur),1))
for k,v in sorted(c.items()): print(f"  {k:28s} {v[1]/v[0]:16.1f}")
PY
