# end-to-end legs only, N fresh processes per setting: spread of the file-fed rate
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
for A in 1 0; do for i in 1 2 3 4; do
  BENCH_AFFINITY=$A timeout 200 python bench.py --full-line --no-cpu-baseline --no-side-legs --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('affinity $A', round(d['value']), round(d['e2e_pipeline']['value']), round(d['e2e_pipeline_roi_decode']['value']), round(d['e2e_pipeline_decoder_cache']['value']), d['e2e_pipeline']['cpus_busy'], d['e2e_pipeline']['host_ms_per_operator']['Reader'])"
done; done
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8; nproc; lscpu | grep -E "NUMA|Socket|Model name" | head -8
