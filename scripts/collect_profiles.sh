#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline block refers to.  Run on the GPU box from the repo root:
#   bash scripts/collect_profiles.sh <tag> [bf16x3|bf16|fp32]        (writes gpurun_out/prof_<tag>/..., then scripts/summarise_profiles.py)
# Counter passes are separate runs with --kernel-trace only (never combined with sys/hip/hsa traces).
set -e
TAG=${1:-r02}
MODE=${2:-bf16x3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD=${FG_PROFILE_CMD:-"python3 $ROOT/bench.py --dtype $MODE --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"}   # FG_PROFILE_CMD: any other program of the repo (python3 <abs path> ...), e.g. scripts/dit_bench.py
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc$i -o pmc -- $CMD > $OUT/pmc$i.log 2>&1
  echo "pass $i done: $C"
done
echo "now run: python scripts/summarise_profiles.py gpurun_out/prof_$TAG $TAG   (in the repo; profiles/ does not travel back from the GPU box)"
