#!/bin/bash
# GPU box: N fresh processes of the headline leg alone with per-step return times; prints value and the largest gaps.
#   bash tools/outlier_hunt.sh N [bench args]
N=${1:-30}; shift
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 $N); do
  BENCH_STEP_TIMES=1 timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-e2e --no-cpu-baseline "$@" > /tmp/oh.json 2> /tmp/oh.err
  python - <<'PY'
import json
d = json.loads(open("/tmp/oh.json").read().strip().splitlines()[-1])
ln = [l for l in open("/tmp/oh.err") if "step return times" in l]
t = [float(x) for x in ln[0].split(":")[1].split()] if ln else []
g = [round(b - a, 2) for a, b in zip([0.0] + t, t)]
print(round(d["value"]), round(d["ms_per_step"], 3), "host", d["config"]["host_ms_per_step"], "gaps", g if d["value"] < 450000 else sorted(g)[-2:])
PY
done
