#!/bin/bash
# where the time of SyncKernel / IndexedSyncKernel goes, workgroup by workgroup (stamps variant of the library)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_m
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
cp dali_amd/lib/libdali_amd_kernels.so /tmp/main_kernels.so
cp build_variants/libdali_amd_kernels_stamps.so dali_amd/lib/libdali_amd_kernels.so
timeout 300 python tools/stamp_probe.py $OUT 2>&1 | tail -40
cp /tmp/main_kernels.so dali_amd/lib/libdali_amd_kernels.so
