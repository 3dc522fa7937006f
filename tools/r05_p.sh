#!/bin/bash
# SyncKernel with the overflow lanes of the write phase packed into the first waves: stamps inside the bench, bench A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_p
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
cp dali_amd/lib/libdali_amd_kernels.so /tmp/main_kernels.so
cp build_variants/libdali_amd_kernels_stamps.so dali_amd/lib/libdali_amd_kernels.so
timeout 300 python tools/stamp_bench.py $OUT/stamps --steps 60 --warmup 8 --no-e2e --no-cpu-baseline --no-side-legs --inflight 1 2>&1 >$OUT/stamps.json | grep -v amdgpu.ids | grep -E "sync|col +(1|2|3|11|12):|start|CUs"
cp /tmp/main_kernels.so dali_amd/lib/libdali_amd_kernels.so
timeout 600 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_jpeg_index.py tests/test_gpu_encoded_cache.py tests/test_gpu_headline.py tests/test_gpu_roi_fusion.py -m gpu -q -x 2>&1 | tail -2
INFLIGHTS="1 5" VARIANT_TESTS=tests/test_gpu_jpeg.py BENCH_ARGS="--steps 200 --warmup 8 --no-e2e --no-cpu-baseline --no-side-legs" bash tools/gpu_lib_variants.sh r05_p main
for i in 1 2; do timeout 300 python bench.py --no-e2e --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-cmd', round(d['value']), d['ms_per_step'], {k: round(v['avg_ms'],3) for k,v in d['roofline']['per_kernel'].items()})"; done
