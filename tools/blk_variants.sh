#!/bin/bash
# experiment: BlockKernel workgroup shapes (waves per workgroup x tasks per wave), headline bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/dali_amd/csrc
for V in ${VARIANTS:-"2 3" "3 2" "12 1" "4 3" "6 2"}; do
  set -- $V
  touch jpeg_huffman.hip; make CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I../../include -Wno-unused-function -DDALIAMD_BLOCK_WAVES=$1 -DDALIAMD_BLOCK_TASKS=$2" > /dev/null 2>&1
  echo "variant waves=$1 tasks=$2"
  for ARGS in "--inflight 1" "--inflight 2" "--inflight 2" "--inflight 2 --batch 512 --batches 2"; do
  (cd $R && BENCH_SKIP_SELF_CHECK=1 timeout 300 python bench.py --full-line $ARGS --no-e2e --no-cpu-baseline --steps 60 --warmup 6 2>/tmp/blk_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$ARGS', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['per_kernel']['BlockKernel']['avg_ms'],4))")
  done
done
