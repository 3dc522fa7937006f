#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
DALI_AMD_OUTPUT_WAIT=poll timeout 300 python tools/e2e_only.py "E2E_WORLD8=1" "E2E_WORLD8=0" 2>&1 | grep -v amdgpu.ids | grep "img/s"
