#!/bin/bash
# Round 5: host side (header cache, streaming copies), layouts / per-sample filters on hardware.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_e
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_header_cache.py tests/test_gpu_resize_layouts.py tests/test_gpu_roi_resize.py tests/test_gpu_pipeline.py tests/test_gpu_augment.py -x -q ) > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for i in 1 2 3; do
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/run$i.json 2> $OUT/run$i.err
  python - $OUT/run$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]
print(round(d["value"]), round(d["ms_per_step"], 4), {k: (round(v) if isinstance(v, float) and v > 100 else v) for k, v in c.items() if k.endswith("_per_s")})
for k in ("e2e_pipeline", "e2e_pipeline_local_world8"):
    e = d.get(k, {})
    print("  ", k, round(e.get("value", 0)), e.get("cpu_ms_per_batch"), e.get("cpu_ms_per_batch_by_thread_group"))
print("   heavy_aug", d.get("heavy_aug", {}).get("value"), "audio", d.get("audio", {}).get("value"))
PY
done
