export PYTHONPATH=$PWD ATTN_SHAPES=${ATTN_SHAPES:-2}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for W in ${FA_PMC_W:-0 1}; do
export FASTGEN_AMD_FA_WIDE=$W
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
i=$((i+1))
timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/fapmc_${W}_$i -- python3 scripts/attn_bench.py > gpurun_out/fapmc_${W}_$i.log 2>&1
python3 scripts/attn_bench.py --pmc gpurun_out/fapmc_${W}_$i
done
done
