#!/bin/bash
# Round 5, after the last host-side change (the executor's waits ask and sleep): the whole gpu test set, five driver-command
# runs and the default bench line again (the rocprof / counter summaries of tools/r05_final_c.sh stay: no kernel changed).
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
w = d.get("e2e_pipeline_local_world8", {})
print(round(d["value"]), round(d["ms_per_step"], 4), {k.replace("_images_per_s", ""): (round(v) if isinstance(v, float) and v > 100 else v) for k, v in c.items() if k.endswith("_per_s")}, "pcie", round(c.get("e2e_pcie_frac") or 0, 3), "heavy", round(d["heavy_aug"]["value"]), "audio", round(d["audio"]["value"]), "e2e cpu", d["e2e_pipeline"]["cpu_ms_per_batch"], "world8 cpu", w.get("cpu_ms_per_batch"))
PY
done
echo "five runs: ${SECONDS}s"
( time timeout 600 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default bench: ${SECONDS}s"
