#!/bin/bash
# experiment: resample kernel variants timed with the headline bench (per-kernel event timings, inflight 1)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/dali_amd/csrc
for V in ${VARIANTS:-0 1 2 3}; do
  touch resample.hip; make CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I../../include -Wno-unused-function -DRS_EXP=$V" > /dev/null 2>&1
  echo "variant $V"
  (cd $R && BENCH_SKIP_SELF_CHECK=1 python bench.py --inflight 1 --no-e2e --no-cpu-baseline --steps 40 --warmup 4 2>/tmp/rs_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:round(v['avg_ms'],4) for k,v in d['roofline']['per_kernel'].items()})")
  tail -3 /tmp/rs_err.log
done
