#!/bin/bash
# gemm_bf16_w4_kernel timing experiments: one library per W4_EXP bit mask (1 no staging at all, 2 no fragment reads, 4 every K-step
# re-reads the first one's operands (cache-resident), 8 no LDS writes).  Results of these builds are wrong; only the time counts.
# Build HERE (hipcc), run on the GPU box:  bash scripts/w4_exp.sh build "1 4 8"   |   bash scripts/w4_exp.sh run "1 4 8"
cd "$(dirname "$0")/.."
C=fastgen_amd/csrc
if [ "$1" = build ]; then
  mkdir -p gpurun_x
  for X in $2; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Xclang -target-feature -Xclang -packed-fp32-ops \
        -DW4_EXP=$X -c $C/gemm.hip -o /tmp/gemm_x$X.o 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -o gpurun_x/libw4_x$X.so \
        $(for o in conv conv_ws conv_ws3 dit wan attn misc aux wgrad bwd attn_bwd disc engine; do echo $C/$o.o; done) /tmp/gemm_x$X.o
  done
  exit 0
fi
export PYTHONPATH=$PWD
for X in $2; do
  echo "W4_EXP=$X"
  FA_LIB=gpurun_x/libw4_x$X.so timeout -k 10 200 python3 scripts/gemm_bench.py --order=513 2>&1 | grep order
done
