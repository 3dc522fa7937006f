#!/bin/bash
# GPU box: N fresh processes of the configs[2] leg; value, step, per-kernel in-schedule / alone ms.   bash tools/aug_runs.sh TAG N [ENV...]
TAG=${1:-aug}; N=${2:-6}; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for i in $(seq 1 $N); do
  env "$@" timeout 200 python bench.py --workload heavy_aug --steps 100 --warmup 8 --no-cpu-baseline > $OUT/run$i.json 2> $OUT/run$i.err
  python - $OUT/run$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
pk = d["roofline"]["per_kernel"]
print(round(d["value"]), round(d["ms_per_step"], 4), {k.replace("Kernel", ""): (round(v["in_schedule_ms"], 3), round(v["avg_ms"], 3)) for k, v in pk.items()}, "host", round(d["config"]["host_ms_per_step"], 3))
PY
done
