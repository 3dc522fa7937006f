#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's numbers on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of the default bench command                       -> gpurun_out/prof_<tag>/stats
#   2. FETCH_SIZE and WRITE_SIZE in two separate --pmc passes (MI355X_MICROARCH.md: they do not fit in one pass;
#      no sys/hip/hsa trace domains together with --pmc)                      -> gpurun_out/prof_<tag>/pmc_*
# then tools/summarize_profiles.py turns the CSVs into the small files committed under profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-e2e"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --inflight 1 > /dev/null 2> $OUT/pmc_$C.log
done
python $R/tools/summarize_profiles.py $OUT $TAG
