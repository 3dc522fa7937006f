#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_l
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_jpeg_index.py tests/test_gpu_encoded_cache.py tests/test_gpu_headline.py -q -x 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --cache-type indexed --steps 200 --no-e2e --no-cpu-baseline > $OUT/i$i.json 2> $OUT/i$i.err
python - $OUT/i$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"], 4), d["config"]["host_ms_per_step"], "in-schedule", {k: round(v["avg_ms"], 3) for k, v in d["roofline"]["per_kernel"].items()})
print("   alone", {k: round(v, 4) for k, v in d["config"]["pipeline"]["single_stream_kernel_ms"].items()}, d["config"]["pipeline"]["encoded_cache"])
PY
done
