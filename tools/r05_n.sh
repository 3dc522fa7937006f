#!/bin/bash
# wave-cooperative write-out of the block-start lists (SyncKernel) and of the per-block records (IndexedSyncKernel):
# list caps 17 / 33 / 65, stamps of the new code
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_n
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
cp dali_amd/lib/libdali_amd_kernels.so /tmp/main_kernels.so
for V in stamps stamps33; do
  cp build_variants/libdali_amd_kernels_$V.so dali_amd/lib/libdali_amd_kernels.so
  mkdir -p $OUT/$V
  timeout 300 python tools/stamp_probe.py $OUT/$V 2>&1 | grep -v amdgpu.ids | tail -30
done
cp /tmp/main_kernels.so dali_amd/lib/libdali_amd_kernels.so
timeout 600 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_jpeg_index.py tests/test_gpu_encoded_cache.py -m gpu -q -x 2>&1 | tail -2
INFLIGHTS="1 5" VARIANT_TESTS=tests/test_gpu_jpeg.py BENCH_ARGS="--steps 200 --warmup 8 --no-e2e --no-cpu-baseline --no-side-legs" bash tools/gpu_lib_variants.sh r05_n main cap33 cap65
timeout 300 python bench.py --cache-type indexed --steps 200 --no-e2e --no-cpu-baseline --no-side-legs > $OUT/indexed.json 2> $OUT/indexed.err
python - $OUT/indexed.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("indexed", round(d["value"]), round(d["ms_per_step"], 4), "in-schedule", {k: round(v["avg_ms"], 3) for k, v in d["roofline"]["per_kernel"].items()})
print("   alone", {k: round(v, 4) for k, v in d["config"]["pipeline"]["single_stream_kernel_ms"].items()})
PY
