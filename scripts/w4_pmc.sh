export PYTHONPATH=$PWD
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
i=$((i+1))
rm -rf gpurun_out/w4pmc
timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/w4pmc -- python3 scripts/gemm_bench.py --order=513 --order=33 > gpurun_out/w4pmc.log 2>&1
python3 - <<PY
import csv, glob
from collections import defaultdict
acc, cnt = defaultdict(float), defaultdict(int)
for f in glob.glob("gpurun_out/w4pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm_bf16_w4" in n or "gemm_bf16_pp_kernel" in n:
            key = ("w4" if "w4" in n else "pp", r["Counter_Name"]); acc[key] += float(r["Counter_Value"]); cnt[key] += 1
for key in sorted(acc): print(key[0], key[1].ljust(28), "%16.0f" % (acc[key] / cnt[key]), cnt[key])
PY
done
rm -rf gpurun_out/w4pmc
