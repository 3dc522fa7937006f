#!/bin/bash
# One gpurun call: GPU test suite + bench variants; everything lands in gpurun_out/<tag>/.
TAG=${1:-chk}
MODE=${2:-full}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
# the decoder tests first, under a short timeout: a hung kernel must not eat the box
( time timeout 300 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_config1.py -x -q ) > $OUT/pytest_jpeg.log 2>&1
RC=$?
tail -25 $OUT/pytest_jpeg.log
if [ $RC -ne 0 ]; then echo "decoder tests failed (rc $RC): stopping"; exit 1; fi
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
if [ "$MODE" = "tests" ]; then exit 0; fi
( time timeout 600 python bench.py --full-line ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.err
for IF in 1 4; do
  timeout 300 python bench.py --full-line --inflight $IF --no-e2e --no-cpu-baseline > $OUT/bench_inflight$IF.json 2> $OUT/bench_inflight$IF.err
done
timeout 300 python bench.py --full-line --batch 512 --batches 2 --inflight 2 --no-e2e --no-cpu-baseline > $OUT/bench_b512.json 2> $OUT/bench_b512.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        pk={k:round(v["avg_ms"],4) for k,v in d["roofline"]["per_kernel"].items()}
        print(f.split("/")[-1], round(d["value"]), d["ms_per_step"], d["config"].get("host_ms_per_step"), pk)
        for k in ("e2e_pipeline","e2e_pipeline_roi_decode","e2e_pipeline_decoder_cache","cpu_baseline","cpu_baseline_pillow"):
            if k in d: print("   ",k, round(d[k]["value"]))
    except Exception as e:
        print(f, "unparsed", e)
PY
