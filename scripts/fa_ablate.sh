#!/bin/bash
# Timing builds of fa_kernel<128> that skip one ingredient of the loop each (FASTGEN_AMD_FA_ABL bit mask: 1 exponentials, 2 fragment
# reads, 4 LDS-DMA, 8 hand-over barrier, 16 the O += V P MFMAs), and the register-prefetch variant.  Run on the GPU box from the repo root.
export PYTHONPATH=$PWD
# the switches exist only in the timing library: build it first (`make -C fastgen_amd/csrc timing`); attn_bench.py loads it when FA_TIMING_LIB=1
export FA_TIMING_LIB=1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() {
  rm -rf gpurun_out/attn
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/attn -- python3 scripts/attn_bench.py > gpurun_out/attn_run.log 2>&1
  python3 scripts/attn_bench.py --parse gpurun_out/attn | grep "chunk 0\|chunk 6\|XL/2 B=256"
}
for A in ${FA_ABL_LIST:-0 1 2 4 8 16 3 31}; do echo "ABL=$A"; FASTGEN_AMD_FA_ABL=$A run; done
echo "two-deep register prefetch"; FASTGEN_AMD_FA_DMA=0 run
