#!/bin/bash
# Round 5: resample kernel changes - tests, single-stream time, SQ counters of the headline's kernels.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_h
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_resample.py tests/test_gpu_config1.py tests/test_gpu_roi_fusion.py tests/test_gpu_roi_resize.py tests/test_gpu_resize_layouts.py tests/test_gpu_headline.py tests/test_gpu_cmn.py -x -q ) > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for i in 1 2; do
timeout 300 python bench.py --steps 200 --no-e2e --no-cpu-baseline > $OUT/b$i.json 2> $OUT/b$i.err
python - $OUT/b$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"], 4), "in-schedule", {k: round(v["avg_ms"], 3) for k, v in d["roofline"]["per_kernel"].items()})
print("   alone", {k: round(v, 4) for k, v in d["config"]["pipeline"]["single_stream_kernel_ms"].items()})
PY
done
bash tools/pmc_pass.sh r05_h/pmc "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" > $OUT/pmc.log 2>&1
grep -E "ResampleKernel|BlockKernel|JpegColor|SyncKernel" $OUT/pmc.log | head -12
