#!/bin/bash
# experiment: resample kernel variants timed with the headline bench (per-kernel event timings, inflight 1)
#   VARIANTS="flags1;flags2" bash tools/rs_variants.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/dali_amd/csrc
IFS=';' read -ra VS <<< "${VARIANTS:--DRS_EXP=0}"
for V in "${VS[@]}"; do
  touch resample.hip; make CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I../../include -Wno-unused-function $V" > /dev/null 2>&1
  echo "variant $V"
  (cd $R && BENCH_SKIP_SELF_CHECK=1 python bench.py --full-line --inflight 1 --no-e2e --no-cpu-baseline --steps 40 --warmup 4 2>/tmp/rs_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:round(v['avg_ms'],4) for k,v in d['roofline']['per_kernel'].items() if 'Resample' in k or 'Color' in k})")
  grep -v amdgpu.ids /tmp/rs_err.log | tail -3
done
