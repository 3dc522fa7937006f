#!/bin/bash
# Round 5: executor knobs on the resident headline - compute streams, side stream.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_f
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
for S in 3 4 5 2; do for A in 1 0; do for D in 5 7; do
  DALI_AMD_PIPELINE_STREAMS=$S DALI_AMD_AUX_STREAM=$A timeout 200 python bench.py --steps 200 --inflight $D --no-e2e --no-side-legs --no-cpu-baseline > $OUT/s${S}_a${A}_d$D.json 2> $OUT/s${S}_a${A}_d$D.err
  python - $OUT/s${S}_a${A}_d$D.json "streams=$S aux=$A depth=$D" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d["value"]), round(d["ms_per_step"], 4), d["config"]["host_ms_per_step"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done; done; done
