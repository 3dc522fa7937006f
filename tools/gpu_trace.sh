#!/bin/bash
# Kernel traces (start / end / queue of every dispatch) of the resident hot path under different environments.
#   bash tools/gpu_trace.sh TAG INFLIGHT "ENV1" "ENV2" ...
TAG=$1; IF=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for E in "$@"; do
  i=$((i+1))
  [ "$E" = "-" ] && E=""
  env $E timeout 150 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$i -- python $R/bench.py --full-line --steps 40 --warmup 8 --inflight $IF --no-e2e --no-cpu-baseline > $OUT/t$i.json 2> $OUT/t$i.err
  # the trace of the process with the most dispatches
  best=$(ls -S /tmp/tr$i/*/*kernel_trace.csv | head -1)
  python - "$best" $OUT/t${i}_trace.csv.gz <<'PY'
import csv, gzip, sys
with open(sys.argv[1]) as f, gzip.open(sys.argv[2], "wt") as g:
    w = csv.writer(g)
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].replace("daliamd::", "").replace("void ", "").split("(")[0][:40]
        w.writerow([r["Queue_Id"], r["Dispatch_Id"], name, r["Start_Timestamp"], r["End_Timestamp"]])
PY
  echo "trace $i: $E -> $(tail -1 $OUT/t$i.json | cut -c1-80)"
done
