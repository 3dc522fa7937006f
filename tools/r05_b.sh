#!/bin/bash
# Round 5, second GPU call: the indexed resident streams on hardware - tests, the driver's command three times, the default bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_b
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_jpeg_index.py tests/test_gpu_encoded_cache.py tests/test_gpu_jpeg.py tests/test_gpu_roi_fusion.py tests/test_gpu_headline.py -x -q ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
for i in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/run$i.json 2> $OUT/run$i.err
  python - $OUT/run$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]
print(round(d["value"]), round(d["ms_per_step"], 4), {k: (round(v) if isinstance(v, float) and v > 100 else v) for k, v in c.items() if k.endswith("_per_s") or k.endswith("_image") or k.endswith("_ms") or k.endswith("frac")})
ri = c["pipeline"]["resident_indexed"]
print("  indexed in-schedule", {k: round(v, 4) for k, v in ri["kernel_ms_in_schedule"].items()})
print("  indexed alone      ", {k: round(v, 4) for k, v in (ri["kernel_ms_single_stream"] or {}).items()})
print("  full    alone      ", {k: round(v, 4) for k, v in (c["pipeline"].get("single_stream_kernel_ms") or {}).items()})
PY
done
( time timeout 600 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
python - $OUT/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]
print("default:", round(d["value"]), round(d["ms_per_step"], 4), {k: (round(v) if isinstance(v, float) and v > 100 else v) for k, v in c.items() if k.endswith("_per_s") or k.endswith("_image")})
for k in ("heavy_aug", "audio"):
    if k in d: print(k, d[k].get("value"), d[k].get("roofline", {}).get("bound"), d[k].get("roofline", {}).get("frac"), d[k].get("error"))
PY
