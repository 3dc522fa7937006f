#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
for G in 64 256; do DALI_AMD_GATHER_WGS=$G timeout 300 python tools/e2e_only.py "DALI_AMD_READER_ZERO_COPY=1" 2>&1 | grep -v amdgpu.ids | tail -2; done
