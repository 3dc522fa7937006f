# heavy-augmentation kernels: parity tests, then the configs[2] bench (in-schedule and single-stream kernel times)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_augment.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -4
for A in "" "--inflight 1"; do
timeout 300 python bench.py --workload heavy_aug --no-cpu-baseline $A 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$A', round(d['value']), round(d['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in d['roofline']['per_kernel'].items()})"
done
