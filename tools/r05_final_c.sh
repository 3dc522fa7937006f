#!/bin/bash
# Round 5, closing collection on the final HEAD: the whole gpu test set, five driver-command runs, the default bench line,
# the rocprof evidence (tools/collect_profiles.sh) and the SQ counter passes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_final
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
echo "tests: ${SECONDS}s"
for i in 1 2 3 4 5; do
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/drv$i.json 2> $OUT/drv$i.err
  python - $OUT/drv$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]
print(round(d["value"]), round(d["ms_per_step"], 4), {k: (round(v) if isinstance(v, float) and v > 100 else v) for k, v in c.items() if k.endswith("_per_s")}, "pcie", round(c.get("e2e_pcie_frac") or 0, 3), "heavy", round(d["heavy_aug"]["value"]), "audio", round(d["audio"]["value"]), "cpu", round(d["cpu_baseline"]["value"]))
PY
done
echo "five runs: ${SECONDS}s"
( time timeout 600 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default bench: ${SECONDS}s"
bash tools/collect_profiles.sh r05 > $OUT/collect.log 2>&1
tail -3 $OUT/collect.log
echo "profiles: ${SECONDS}s"
bash tools/pmc_pass.sh r05_final/pmc "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" > $OUT/pmc.log 2>&1
BENCH_ARGS="--cache-type indexed --steps 8 --warmup 4 --no-cpu-baseline --no-e2e --inflight 1" bash tools/pmc_pass.sh r05_final/pmc_indexed "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" > $OUT/pmc_indexed.log 2>&1
echo "counters: ${SECONDS}s"
