#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_k
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_formats.py tests/test_gpu_jpeg.py tests/test_gpu_pipeline.py tests/test_gpu_encoded_cache.py tests/test_gpu_config1.py -q -x 2>&1 | tail -2
for i in 1 2; do
for CT in encoded indexed; do
timeout 300 python bench.py --cache-type $CT --steps 200 --no-e2e --no-cpu-baseline --no-side-legs > $OUT/$CT$i.json 2> $OUT/$CT$i.err
python - $OUT/$CT$i.json $CT <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["value"]), round(d["ms_per_step"], 4), d["config"]["host_ms_per_step"], d["config"]["pipeline"]["host_ms_per_operator"])
PY
done; done
