#!/bin/bash
# Round 5: the strip form of the matrix-core blur.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_d
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_augment.py -x -q ) > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for V in "0 0" "1 0" "1 1"; do
  set -- $V
  DALI_AMD_BLUR_MFMA=$1 DALI_AMD_BLUR_FUSION=$2 timeout 300 python bench.py --workload heavy_aug --steps 100 --no-cpu-baseline > $OUT/heavy_mfma$1_fusion$2.json 2> $OUT/heavy_mfma$1_fusion$2.err
  python - $OUT/heavy_mfma$1_fusion$2.json "heavy_aug mfma=$1 fusion=$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["value"]), round(d["ms_per_step"], 4), {k: (round(v["avg_ms"], 4), round(v["in_schedule_ms"], 4)) for k, v in d["roofline"]["per_kernel"].items()}, d["config"]["kernels"])
PY
done
