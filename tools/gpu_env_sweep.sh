#!/bin/bash
# Benches the resident hot path under different runtime environments / batches in flight (no rebuild).
#   bash tools/gpu_env_sweep.sh TAG "ENV1" "ENV2" ...      (an ENV is a string like "GPU_MAX_HW_QUEUES=8 FOO=1"; "-" = none)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
i=0
for E in "$@"; do
  i=$((i+1))
  [ "$E" = "-" ] && E=""
  for IF in ${INFLIGHTS:-4}; do
    env $E timeout 300 python bench.py --full-line --steps ${STEPS:-60} --warmup 8 --inflight $IF --no-e2e --no-cpu-baseline > $OUT/e${i}_inflight$IF.json 2> $OUT/e${i}_inflight$IF.err || tail -3 $OUT/e${i}_inflight$IF.err
  done
  echo "env $i: $E"
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/e*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],4), d["config"].get("host_ms_per_step"))
    except Exception as e:
        print(f, "unparsed", e)
PY
