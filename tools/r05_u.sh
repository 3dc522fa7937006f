#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
for G in 16 32 64 128; do echo "== $G workgroups"; DALI_AMD_GATHER_WGS=$G timeout 100 python tools/probes/gather_probe2.py 2>&1 | grep "other files"; done
timeout 500 python tools/e2e_only.py "DALI_AMD_READER_ZERO_COPY=0" "DALI_AMD_READER_ZERO_COPY=1 DALI_AMD_GATHER_WGS=32" 2>&1 | grep -v amdgpu.ids | tail -4
for G in 16 64; do DALI_AMD_GATHER_WGS=$G timeout 300 python tools/e2e_only.py "DALI_AMD_READER_ZERO_COPY=1" 2>&1 | grep -v amdgpu.ids | tail -2; done
