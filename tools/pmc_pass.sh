#!/bin/bash
# PMC passes over the bench command (one rocprofv3 run per counter set; no trace domains next to --pmc).
#   bash tools/pmc_pass.sh TAG "SET1" "SET2" ...      each SET = space-separated counter names
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "$@"; do
  i=$((i+1))
  timeout ${PROF_TIMEOUT:-300} rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/set$i -- python $R/bench.py --full-line ${BENCH_ARGS:---steps 8 --warmup 4 --no-cpu-baseline --no-e2e --inflight 1} > /dev/null 2> $OUT/set$i.log
done
python - <<PY
import csv, collections, glob, json
summary = collections.defaultdict(dict)
for d in sorted(glob.glob("$OUT/set*")):
    for f in glob.glob(d+"/*/*_counter_collection.csv"):
        rows=list(csv.DictReader(open(f)))
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            acc[r["Kernel_Name"].split("(")[0][-28:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in sorted(acc.items()):
            if "daliamd" in k or "Kernel" in k:
                avg = {c:round(sum(x)/len(x)) for c,x in v.items()}
                print(k, avg, "launches", len(next(iter(v.values()))))
                summary[k.split("::")[-1].split("<")[0]].update(avg)
json.dump({"note": "rocprofv3 --pmc, averages per launch over the bench command (inflight 1); one pass per counter set; "
           "SQ counters are summed over the shader engines as rocprofv3 reports them", "kernels": summary},
          open("$OUT/summary.json", "w"), indent=1, sort_keys=True)
PY
