#!/bin/bash
# Benches prebuilt variants of the kernel library (build_variants/libdali_amd_kernels_<name>.so, built locally so that
# no GPU time goes into compiling): parity tests of the entropy decoder, then the resident hot path.
#   bash tools/gpu_lib_variants.sh TAG name1 name2 ...     ("main" = the library as built)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
cp dali_amd/lib/libdali_amd_kernels.so /tmp/main_kernels.so
for V in "$@"; do
  if [ $V = main ]; then cp /tmp/main_kernels.so dali_amd/lib/libdali_amd_kernels.so; else cp build_variants/libdali_amd_kernels_$V.so dali_amd/lib/libdali_amd_kernels.so; fi
  timeout 200 python -m pytest ${VARIANT_TESTS:-tests/test_gpu_jpeg.py} -m gpu -q -x 2>&1 | tail -1
  for IF in ${INFLIGHTS:-1 5}; do
    timeout 200 python bench.py --full-line ${BENCH_ARGS:---steps 400 --warmup 8 --no-e2e --no-cpu-baseline} --inflight $IF > $OUT/${V}_inflight$IF.json 2> $OUT/${V}_inflight$IF.err || tail -3 $OUT/${V}_inflight$IF.err
  done
done
cp /tmp/main_kernels.so dali_amd/lib/libdali_amd_kernels.so
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        pk={k.replace("Kernel",""):round(v["avg_ms"] if isinstance(v,dict) else v,3) for k,v in (d["roofline"].get("per_kernel") or d["config"].get("kernels_ms_per_step") or {}).items()}
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],4), pk)
    except Exception as e:
        print(f, "unparsed", e)
PY
