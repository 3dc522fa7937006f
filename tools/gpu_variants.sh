#!/bin/bash
# Builds the kernel library with different -D settings on the GPU box and benches each (resident hot path only).
#   bash tools/gpu_variants.sh TAG "FLAGS1" "FLAGS2" ...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
i=0
for FL in "$@"; do
  i=$((i+1))
  touch dali_amd/csrc/jpeg_huffman.hip
  make -C dali_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I../../include -Wno-unused-function $FL" > $OUT/build$i.log 2>&1 || { tail -5 $OUT/build$i.log; continue; }
  make -C dali_amd/host > /dev/null 2>&1
  for IF in ${INFLIGHTS:-1 4}; do
    timeout 300 python bench.py --full-line --steps 40 --warmup 5 --inflight $IF --no-e2e --no-cpu-baseline > $OUT/v${i}_inflight$IF.json 2> $OUT/v${i}_inflight$IF.err || tail -3 $OUT/v${i}_inflight$IF.err
  done
  echo "variant $i: $FL"
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/v*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        pk={k[:-6]:round(v["avg_ms"],3) for k,v in d["roofline"]["per_kernel"].items()}
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],4), pk)
    except Exception as e:
        print(f, "unparsed", e)
PY
