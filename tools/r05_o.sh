#!/bin/bash
# why is list cap 33 faster in the stand-alone probe and slower inside the bench?  stamps of the bench's own launches
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_o
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
cp dali_amd/lib/libdali_amd_kernels.so /tmp/main_kernels.so
for V in stamps stamps33; do
  cp build_variants/libdali_amd_kernels_$V.so dali_amd/lib/libdali_amd_kernels.so
  echo "== $V"
  timeout 300 python tools/stamp_bench.py $OUT/$V --steps 60 --warmup 8 --no-e2e --no-cpu-baseline --no-side-legs --inflight 1 2>&1 >$OUT/$V.json | grep -v amdgpu.ids | tail -22
done
cp /tmp/main_kernels.so dali_amd/lib/libdali_amd_kernels.so
