#!/bin/bash
# GPU box: per-kernel durations of the headline graph on data set variants (tools/variant_kernels.py), one and five batches in flight.
#   bash tools/variant_report.sh TAG VARIANT...
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
for V in "$@"; do
  for D in 1 5; do
    python tools/variant_kernels.py $V --inflight $D > $OUT/${V}_d$D.json 2> $OUT/${V}_d$D.err || tail -5 $OUT/${V}_d$D.err
    python - $OUT/${V}_d$D.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"], 3), "host", d["host_ms_per_operator"])
print("   ", {k: (v["launches"], round(v["avg_ms"], 3)) for k, v in d.get("kernel_ms_in_schedule", {}).items()})
PY
  done
done
