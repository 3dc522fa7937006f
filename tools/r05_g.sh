#!/bin/bash
# Round 5: decoder front on the side stream - tests + A/B.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_g
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_encoded_cache.py tests/test_gpu_config1.py tests/test_gpu_roi_fusion.py tests/test_gpu_headline.py tests/test_gpu_decoder_cache.py tests/test_gpu_pipeline.py -x -q ) > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for F in 1 0 1 0; do
  DALI_AMD_DECODER_FRONT_ON_SIDE_STREAM=$F timeout 300 python bench.py --steps 200 --no-e2e --no-side-legs --no-cpu-baseline > $OUT/f$F.json 2> $OUT/f$F.err
  python - $OUT/f$F.json "front_on_side=$F (200 steps)" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["value"]), round(d["ms_per_step"], 4), d["config"]["host_ms_per_step"], {k: round(v["avg_ms"], 3) for k, v in d["roofline"]["per_kernel"].items()})
PY
  DALI_AMD_DECODER_FRONT_ON_SIDE_STREAM=$F timeout 300 python bench.py --steps 20 --warmup 5 --no-side-legs --no-cpu-baseline > $OUT/g$F.json 2> $OUT/g$F.err
  python - $OUT/g$F.json "front_on_side=$F (20 steps, e2e)" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]
print(sys.argv[2], round(d["value"]), {k: (round(v) if isinstance(v, float) and v > 100 else v) for k, v in c.items() if k.endswith("_per_s")})
e = d.get("e2e_pipeline_local_world8", {})
print("   world8 cpu", e.get("cpu_ms_per_batch"), e.get("cpu_ms_per_batch_by_thread_group"))
PY
done
