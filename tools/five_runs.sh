#!/bin/bash
# Five consecutive fresh processes of the command the driver runs; prints value / ms_per_step / host ms of each.
#   bash tools/five_runs.sh TAG
TAG=${1:-five}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for i in 1 2 3 4 5; do
  timeout 240 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/run$i.json 2> $OUT/run$i.err
  python - $OUT/run$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"], 4), d["config"]["host_ms_per_step"])
PY
done
