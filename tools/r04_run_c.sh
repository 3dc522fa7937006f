# fused colour output: tests, then the headline bench with and without it
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r04c}; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_jpeg.py -x -q -k "fused_colour" ) > $OUT/pytest_fused.log 2>&1; tail -15 $OUT/pytest_fused.log
( timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_config1.py tests/test_gpu_pipeline.py tests/test_gpu_encoded_cache.py tests/test_gpu_roi_resize.py -x -q ) > $OUT/pytest_jpeg.log 2>&1; tail -5 $OUT/pytest_jpeg.log
for F in 1 0; do
  DALI_AMD_FUSE_COLOR=$F timeout 300 python bench.py --no-cpu-baseline --no-e2e --steps 100 > $OUT/bench_fuse$F.json 2> $OUT/bench_fuse$F.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        pk={k:round(v["avg_ms"],4) for k,v in d["roofline"]["per_kernel"].items()}
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],4), pk)
        print("    single", {k:round(v,4) for k,v in d["config"]["pipeline"].get("single_stream_kernel_ms",{}).items()})
    except Exception as e:
        print(f, "unparsed", e); print(open(f.replace(".json",".err")).read()[-1500:])
PY
