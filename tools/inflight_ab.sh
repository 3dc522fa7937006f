R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
for rep in 1 2 3; do for IF in 5 6 7; do
  A=$(timeout 200 python bench.py --full-line --no-cpu-baseline --no-e2e --no-side-legs --inflight $IF --steps 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']))")
  B=$(timeout 200 python bench.py --full-line --no-cpu-baseline --no-e2e --no-side-legs --inflight $IF --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']))")
  echo "inflight $IF: 300 steps $A   20 steps $B"
done; done
