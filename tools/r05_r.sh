#!/bin/bash
# zero-copy reader (registered file mappings + device-side fetch): tests, then the end-to-end legs with and without it
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_r
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_gather.py tests/test_gpu_reader_zero_copy.py tests/test_gpu_encoded_cache.py tests/test_gpu_pipeline.py tests/test_gpu_decoder_cache.py -m gpu -q -x 2>&1 | tail -3
for ZC in 1 0; do
  DALI_AMD_READER_ZERO_COPY=$ZC timeout 400 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-side-legs > $OUT/zc$ZC.json 2> $OUT/zc$ZC.err || tail -5 $OUT/zc$ZC.err
  python - $OUT/zc$ZC.json $ZC <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("zero_copy", sys.argv[2], "value", round(d["value"]))
for k in ("e2e_pipeline", "e2e_pipeline_roi_decode", "e2e_pipeline_decoder_cache", "e2e_pipeline_local_world8"):
    e = d.get(k)
    if not e: continue
    print("  %-28s %8.0f img/s  cpu %.2f ms/batch %s  host/dev stage %.3f/%.3f  pcie %s  gather %s" % (
        k, e["value"], e["cpu_ms_per_batch"], e["cpu_ms_per_batch_by_thread_group"], e["host_stage_ms_per_batch"], e["device_stage_ms_per_batch"],
        (round(e["pcie"]["frac"], 3), round(e["pcie"]["peak"], 1)) if "pcie" in e else None, "gather_encoded" in e.get("kernels", [])))
PY
done
