#!/bin/bash
# N consecutive fresh processes of the command the driver runs; value / ms_per_step of each.   bash tools/n_runs.sh TAG N [bench args]
TAG=${1:-nruns}; N=${2:-8}; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for i in $(seq 1 $N); do
  BENCH_STEP_TIMES=${STEP_TIMES:-0} timeout 240 python bench.py --gpus 1 --steps 20 --warmup 5 "$@" > $OUT/run$i.json 2> $OUT/run$i.err
  python - $OUT/run$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]
print(round(d["value"]), round(d["ms_per_step"], 4), "e2e", round(c.get("e2e_images_per_s") or 0), "idx", round(c.get("resident_indexed_images_per_s") or 0),
      "dht", round(c.get("value_distinct_dht") or 0), "mixed", round(c.get("value_mixed") or 0), "large", round(c.get("value_large_images") or 0),
      "4GB", round(c.get("value_resident_4GB") or 0))
PY
  grep "step return times" $OUT/run$i.err | python -c "
import sys
for ln in sys.stdin:
    t = [float(x) for x in ln.split(':')[1].split()]
    g = [b - a for a, b in zip([0.0] + t, t)]
    print('    largest gaps between step returns (ms):', sorted(round(x, 2) for x in g)[-3:], 'median', sorted(g)[len(g) // 2])
"
done
