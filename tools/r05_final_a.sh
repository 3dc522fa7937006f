#!/bin/bash
# Round 5: the whole gpu suite + the rocprof evidence of the state that ships.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_final
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $OUT/pytest_gpu.log 2>&1
tail -4 $OUT/pytest_gpu.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.log 2>&1
tail -2 $OUT/smoke.log
bash tools/collect_profiles.sh r05 > $OUT/collect.log 2>&1
tail -12 $OUT/collect.log
