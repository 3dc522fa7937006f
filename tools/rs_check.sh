#!/bin/bash
# quick check after a resample kernel change: its parity tests, then the headline bench (per-kernel timings)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_resample.py tests/test_gpu_config1.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -8
for ARGS in "--inflight 1" "--inflight 2"; do
  timeout 300 python bench.py --full-line $ARGS --no-e2e --no-cpu-baseline --steps 60 --warmup 6 2>/tmp/rs_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$ARGS', round(d['value']), round(d['ms_per_step'],4), {k[:-6]:round(v['avg_ms'],4) for k,v in d['roofline']['per_kernel'].items()})"
  tail -2 /tmp/rs_err.log
done
