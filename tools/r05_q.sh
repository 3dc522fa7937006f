#!/bin/bash
# BlockKernel: start / end of every workgroup inside the bench (one batch in flight)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_q
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
cp dali_amd/lib/libdali_amd_kernels.so /tmp/main_kernels.so
cp build_variants/libdali_amd_kernels_stamps.so dali_amd/lib/libdali_amd_kernels.so
timeout 300 python tools/stamp_bench.py $OUT/stamps --steps 60 --warmup 8 --no-e2e --no-cpu-baseline --no-side-legs --inflight 1 2>&1 >$OUT/stamps.json | grep -v amdgpu.ids | grep -A12 "^block"
cp /tmp/main_kernels.so dali_amd/lib/libdali_amd_kernels.so
