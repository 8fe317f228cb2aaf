export PYTHONPATH=$PWD
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for A in 0 1 2 4 8 16 3 31; do
  rm -rf gpurun_out/attn
  FASTGEN_AMD_FA_ABL=$A timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/attn -- python3 scripts/attn_bench.py > gpurun_out/attn_run.log 2>&1
  echo "ABL=$A"; python3 scripts/attn_bench.py --parse gpurun_out/attn | grep "chunk 6\|XL/2 B=256"
done
