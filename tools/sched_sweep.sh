# headline bench over batches in flight x compute streams (resident hot path only, 200-step regions)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
for S in 3 2 4; do for IF in 4 5 6 7; do
  V=$(DALI_AMD_PIPELINE_STREAMS=$S timeout 200 python bench.py --full-line --no-cpu-baseline --no-e2e --no-side-legs --inflight $IF --steps 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']))")
  echo "streams $S inflight $IF: $V"
done; done
