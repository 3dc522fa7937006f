#!/bin/bash
# rocprofv3 kernel stats + PMC passes of the audio side bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-audio_prof}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --full-line --workload ${WORKLOAD:-audio} --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/stats.log
python - <<PY
import csv,glob
for f in glob.glob("$OUT/stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
cd $R
BENCH_ARGS="--workload ${WORKLOAD:-audio} --steps 5 --warmup 2" bash tools/pmc_pass.sh $TAG/pmc "$@"
