#!/bin/bash
# The kernel-trace + stats pass of tools/collect_profiles.sh alone (headline workload), with every launch of the process at
# the full depth (--no-side-legs): the table whose per-kernel averages must agree with bench.py's live durations.
#   bash tools/collect_stats_only.sh TAG
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/stats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout ${PROF_TIMEOUT:-120} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --full-line --steps 40 --warmup 3 --no-cpu-baseline --no-e2e --no-side-legs > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
best=$(ls -S $OUT/stats/*/*kernel_stats.csv | head -1)
cp "$best" $OUT/${TAG}_kernel_stats_full_depth.csv
head -12 "$best" | cut -c1-160
python - $OUT/bench_under_rocprof.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench under rocprof:", round(d["value"]), {k: round(v["avg_ms"], 4) for k, v in d["roofline"]["per_kernel"].items()})
PY
