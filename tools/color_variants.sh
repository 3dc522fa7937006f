#!/bin/bash
# experiment: rows per thread of the 4:2:0 colour path (headline bench, inflight 1 and 2)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/dali_amd/csrc
for V in ${VARIANTS:-8 4 12 16}; do
  touch jpeg_color.hip; make CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I../../include -Wno-unused-function -DDALIAMD_COLOR_ROWS=$V" > /dev/null 2>&1
  echo "rows $V"
  for IF in 1 2; do
  (cd $R && timeout 300 python bench.py --full-line --inflight $IF --no-e2e --no-cpu-baseline --steps 60 --warmup 6 2>/tmp/c_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' inflight $IF', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['per_kernel']['JpegColorKernel']['avg_ms'],4))")
  done
done
