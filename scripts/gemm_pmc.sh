#!/bin/bash
# Counters of the hand-written token GEMM and of hipBLASLt (torch.matmul) on the same shapes, one process: which one keeps the matrix
# pipe busier, at which clock, with how many instructions.  Run on the GPU box from the repo root; prints per-kernel averages.
export PYTHONPATH=$PWD
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rm -rf gpurun_out/gemmpmc_$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/gemmpmc_$i -- python3 scripts/gemm_bench.py --order=17 > gpurun_out/gemmpmc_$i.log 2>&1
  python3 - <<PY
import csv, glob
from collections import defaultdict
acc, cnt = defaultdict(float), defaultdict(int)
dur, dc = defaultdict(float), defaultdict(int)
for f in glob.glob("gpurun_out/gemmpmc_$i/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm_bf16_pp" in n or "Cijk" in n:
            key = (n[:60], r["Grid_Size"], r["Counter_Name"])
            acc[key] += float(r["Counter_Value"]); cnt[key] += 1
for key in sorted(acc):
    print(f"{key[0]:60s} grid {key[1]:>9s} {key[2]:28s} {acc[key] / cnt[key]:16.0f}  (n={cnt[key]})")
PY
done
