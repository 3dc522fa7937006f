#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
env | grep -E "^(HSA|HIP|ROC|GPU_|AMD|OMP|KMP|MKL)" | head -20
E2E_THREADS=1 timeout 300 python tools/e2e_only.py "E2E_WORLD8=1" 2>&1 | grep -v amdgpu.ids | tail -12
