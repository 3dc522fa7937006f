cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
for G in 0 1; do
  BENCH_QUIET_GC=$G timeout 240 python bench.py --gpus 1 --steps 20 --warmup 5 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gc_quiet=$G', round(d['value']), round(d['ms_per_step'], 4))"
done; done
