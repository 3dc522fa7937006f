R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
for A in 0 1 0 1; do
  BENCH_AFFINITY=$A timeout 200 python bench.py --no-cpu-baseline --no-e2e --no-side-legs --steps 200 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('resident affinity $A', round(d['value']), d['config']['host_ms_per_step'])"
  BENCH_AFFINITY=$A timeout 200 python bench.py --no-cpu-baseline --no-e2e --no-side-legs --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   20 steps', round(d['value']))"
done
