#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's numbers on the GPU box (run through gpurun from the repo root):
#   per workload (headline, heavy_aug = configs[2], audio = configs[3]):
#   1. kernel trace + stats of the bench command                              -> gpurun_out/prof_<tag>/<workload>/stats
#   2. FETCH_SIZE and WRITE_SIZE in two separate --pmc passes (MI355X_MICROARCH.md: they do not fit in one pass;
#      no sys/hip/hsa trace domains together with --pmc)                      -> gpurun_out/prof_<tag>/<workload>/pmc_*
#   3. the plain bench line (no profiler attached)                            -> gpurun_out/<tag>_summary/<tag>_<workload>_bench.json
# then tools/summarize_profiles.py turns the CSVs into the small files committed under profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
SUM=$R/gpurun_out/${TAG}_summary
mkdir -p $SUM
cd /tmp && export TMPDIR=/tmp
for W in ${WORKLOADS:-headline indexed heavy_aug audio normalize}; do
  OUT=$R/gpurun_out/prof_$TAG/$W
  mkdir -p $OUT
  CMD="python $R/bench.py"
  if [ $W = headline ]; then
    ARGS="--steps 40 --warmup 3 --no-cpu-baseline --no-e2e --no-side-legs"; PMCARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-side-legs --inflight 1"; SUF=""
  elif [ $W = indexed ]; then     # the headline graph on streams resident WITH their side information (cache_type="indexed")
    ARGS="--cache-type indexed --steps 40 --warmup 3 --no-cpu-baseline --no-e2e --no-side-legs"; PMCARGS="--cache-type indexed --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-side-legs --inflight 1"; SUF="_indexed"
  elif [ $W = normalize ]; then   # fn.normalize (wave64 mean / stddev reductions) + the stand-alone CropMirrorNormalize kernel
    CMD="python $R/tools/normalize_prof.py"; ARGS="20"; PMCARGS="5"; SUF="_$W"
  else
    ARGS="--workload $W --steps 20 --warmup 3 --no-cpu-baseline"; PMCARGS="--workload $W --steps 5 --warmup 1 --no-cpu-baseline"; SUF="_$W"
  fi
  timeout ${PROF_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout ${PROF_TIMEOUT:-300} rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- $CMD $PMCARGS > /dev/null 2> $OUT/pmc_$C.log
  done
  python $R/tools/summarize_profiles.py $OUT $TAG$SUF $SUM || { echo "collect_profiles: $W summary FAILED"; tail -5 $OUT/*.log; FAILED=1; }
  if [ $W = normalize ]; then
    (cd $R && timeout 300 python tools/normalize_prof.py 2>/dev/null | tail -1 > $SUM/${TAG}_${W}_bench.json)
  elif [ $W = indexed ]; then
    (cd $R && timeout 300 python bench.py --full-line --cache-type indexed --no-e2e --no-cpu-baseline 2>/dev/null | tail -1 > $SUM/${TAG}_${W}_bench.json)
  elif [ $W != headline ]; then
    (cd $R && timeout 300 python bench.py --full-line --workload $W 2>/dev/null | tail -1 > $SUM/${TAG}_${W}_bench.json)
  fi
done
ls -la $SUM
exit ${FAILED:-0}
