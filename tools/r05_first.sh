#!/bin/bash
# Round 5, first GPU contact of the ROI-fused headline: tests, A/B of DALI_AMD_ROI_FUSION over five driver-command
# runs each, rocprof stats + traffic of the configuration `value` runs.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_first
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 400 python -m pytest tests/test_gpu_roi_fusion.py tests/test_gpu_headline.py -x -q ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
for F in 1 0; do
  echo "DALI_AMD_ROI_FUSION=$F"
  DALI_AMD_ROI_FUSION=$F bash tools/five_runs.sh r05_first/fusion$F
done
WORKLOADS=headline bash tools/collect_profiles.sh r05a > $OUT/collect.log 2>&1
tail -5 $OUT/collect.log
