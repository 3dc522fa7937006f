#!/bin/bash
# Round 5, third GPU call: host-built code tables + side stream for the resampling tables, the blur on the matrix cores.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_c
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_augment.py tests/test_gpu_jpeg_index.py tests/test_gpu_encoded_cache.py tests/test_gpu_config1.py tests/test_gpu_roi_fusion.py tests/test_gpu_resample.py tests/test_gpu_jpeg.py tests/test_gpu_pipeline.py -x -q ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
show() {
python - "$@" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
if "heavy" in d["metric"]:
    print(sys.argv[2], round(d["value"]), round(d["ms_per_step"], 4), {k: (round(v["avg_ms"], 4), round(v["in_schedule_ms"], 4)) for k, v in d["roofline"]["per_kernel"].items()}, d["config"]["kernels"])
else:
    c = d["config"]
    print(sys.argv[2], round(d["value"]), round(d["ms_per_step"], 4), {k: (round(v) if isinstance(v, float) and v > 100 else v) for k, v in c.items() if k.endswith("_per_s") or k.endswith("_image") or k.endswith("_ms")})
    ri = c["pipeline"].get("resident_indexed")
    if ri:
        print("  indexed in-schedule", {k: round(v, 4) for k, v in ri["kernel_ms_in_schedule"].items()}, ri["ms_per_step"])
        print("  indexed alone      ", {k: round(v, 4) for k, v in (ri["kernel_ms_single_stream"] or {}).items()})
    print("  value in-schedule  ", {k: round(v["avg_ms"], 4) for k, v in d["roofline"]["per_kernel"].items()})
    print("  host", c["pipeline"]["host_ms_per_operator"], c["pipeline"]["device_stage_ms_per_step"])
PY
}
for V in "0 0" "1 0" "1 1"; do
  set -- $V
  DALI_AMD_BLUR_MFMA=$1 DALI_AMD_BLUR_FUSION=$2 timeout 300 python bench.py --workload heavy_aug --steps 100 > $OUT/heavy_mfma$1_fusion$2.json 2> $OUT/heavy_mfma$1_fusion$2.err
  show $OUT/heavy_mfma$1_fusion$2.json "heavy_aug mfma=$1 fusion=$2"
done
for i in 1 2; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/run$i.json 2> $OUT/run$i.err
  show $OUT/run$i.json "driver-cmd $i"
done
DALI_AMD_TRACE=1 timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default_trace.err
show $OUT/bench_default.json "default(200 steps)"
grep -E "trace\]" $OUT/bench_default_trace.err | head -150 > $OUT/trace.txt
