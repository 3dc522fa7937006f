#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_j
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_roi_fusion.py tests/test_gpu_headline.py tests/test_gpu_config1.py -q -x 2>&1 | tail -2
for i in 1 2; do
for CT in encoded indexed; do
timeout 300 python bench.py --cache-type $CT --steps 200 --no-e2e --no-cpu-baseline --no-side-legs > $OUT/$CT$i.json 2> $OUT/$CT$i.err
python - $OUT/$CT$i.json $CT <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["value"]), round(d["ms_per_step"], 4), d["config"]["host_ms_per_step"], d["config"]["pipeline"]["launches_timed"])
print("   ", {k: round(v["avg_ms"], 3) for k, v in d["roofline"]["per_kernel"].items()})
PY
done; done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs > $OUT/drv.json 2> $OUT/drv.err
python - $OUT/drv.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]
print("driver-cmd", round(d["value"]), {k: (round(v) if isinstance(v, float) and v > 100 else v) for k, v in c.items() if k.endswith("_per_s")})
PY
