set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04b; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_pipeline.py -x -q ) > $OUT/pytest_pipeline.log 2>&1; tail -3 $OUT/pytest_pipeline.log
( timeout 600 python bench.py --no-cpu-baseline ) > $OUT/bench_default.json 2> $OUT/bench_default.err
for T in 6 12 16; do DALI_AMD_READER_THREADS=$T timeout 300 python bench.py --no-cpu-baseline --no-side-legs --steps 100 > $OUT/bench_rd$T.json 2> $OUT/bench_rd$T.err; done
DALI_AMD_READER_MMAP_MB=0 timeout 300 python bench.py --no-cpu-baseline --no-side-legs --steps 100 > $OUT/bench_nommap.json 2> $OUT/bench_nommap.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "iter", round(d.get("iterator",{}).get("value",0)))
        for k in ("e2e_pipeline","e2e_pipeline_roi_decode","e2e_pipeline_decoder_cache","e2e_pipeline_local_world8"):
            if k in d: print("   ",k, round(d[k]["value"]), d[k]["host_ms_per_operator"].get("Reader"), d[k]["cpu_ms_per_batch_by_thread_group"])
    except Exception as e:
        print(f, "unparsed", e)
PY
