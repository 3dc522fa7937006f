#!/bin/bash
# Round 5: IndexedSyncKernel experiments (prebuilt variants) on the indexed headline.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_jpeg_index.py tests/test_gpu_encoded_cache.py -q -x 2>&1 | tail -2
VARIANT_TESTS="tests/test_gpu_jpeg_index.py" INFLIGHTS="1 5" BENCH_ARGS="--cache-type indexed --steps 200 --warmup 8 --no-e2e --no-cpu-baseline --no-side-legs" bash tools/gpu_lib_variants.sh r05_i main idx_nostore slice128
