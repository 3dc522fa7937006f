#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
cp dali_amd/lib/libdali_amd_kernels.so /tmp/main_kernels.so
for V in main gather_u1 gather_u8 gather_t64u8 gather_t1024u1; do
  if [ $V = main ]; then cp /tmp/main_kernels.so dali_amd/lib/libdali_amd_kernels.so; else cp build_variants/libdali_amd_kernels_$V.so dali_amd/lib/libdali_amd_kernels.so; fi
  echo "== $V"
  timeout 120 python tools/probes/gather_probe.py 2>&1 | grep -v amdgpu.ids | tail -7
done
cp /tmp/main_kernels.so dali_amd/lib/libdali_amd_kernels.so
