#!/bin/bash
# fa2_kernel timing experiments: one library per FA2_EXP bit mask (1 barrier every other tile, 2 no LDS-DMA, 4 no fragment re-reads,
# 8 no maximum / reference check, 16 no exponentials, 32 no O^T MFMAs, 64 no chain MFMAs; fa72_seq_kernel: 256 no K / V load,
# 512 one key tile instead of eight).  Results of these builds are garbage; only the
# kernel time counts.  Build HERE (hipcc), run on the GPU box:  bash scripts/fa2_exp.sh build "0 1 2 ..."   |   bash scripts/fa2_exp.sh run "0 1 2 ..."
cd "$(dirname "$0")/.."
C=fastgen_amd/csrc
if [ "$1" = build ]; then
  mkdir -p gpurun_x
  for X in $2; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Xclang -target-feature -Xclang -packed-fp32-ops \
        -DFA2_EXP=$X -c $C/wan.hip -o /tmp/wan_x$X.o 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -o gpurun_x/libfa2_x$X.so \
        $(for o in conv conv_ws conv_ws3 gemm dit attn misc aux wgrad bwd attn_bwd disc engine; do echo $C/$o.o; done) /tmp/wan_x$X.o
  done
  exit 0
fi
export PYTHONPATH=$PWD ATTN_SHAPES=${ATTN_SHAPES:-2}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for X in $2; do
  rm -rf gpurun_out/fa2x
  FA_LIB=gpurun_x/libfa2_x$X.so timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fa2x -- python3 scripts/attn_bench.py > gpurun_out/fa2x.log 2>&1
  echo "FA2_EXP=$X $(python3 scripts/attn_bench.py --parse gpurun_out/fa2x | cut -c1-110)"
done
