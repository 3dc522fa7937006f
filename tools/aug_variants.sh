# configs[2] bench (one batch in flight: kernel cost) for prebuilt variants of the kernel library
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
cp dali_amd/lib/libdali_amd_kernels.so /tmp/main_kernels.so
for V in "$@"; do
  if [ $V = main ]; then cp /tmp/main_kernels.so dali_amd/lib/libdali_amd_kernels.so; else cp build_variants/libdali_amd_kernels_$V.so dali_amd/lib/libdali_amd_kernels.so; fi
  T=$(timeout 300 python -m pytest tests/test_gpu_augment.py -x -q 2>&1 | tail -1)
  timeout 300 python bench.py --full-line --workload heavy_aug --no-cpu-baseline --inflight 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V', '[$T]', round(d['value']), round(d['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in d['roofline']['per_kernel'].items()})"
done
cp /tmp/main_kernels.so dali_amd/lib/libdali_amd_kernels.so
