cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_flat
for F in 0 1; do
  DALI_AMD_INDEX_FLAT_STREAMS=$F python tools/variant_kernels.py flat --inflight 5 --steps 60 > gpurun_out/r06_flat/flat_$F.json 2>/dev/null
  python - gpurun_out/r06_flat/flat_$F.json $F <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("index_flat_streams =", sys.argv[2], round(d["value"]), "img/s", round(d["ms_per_step"], 3), "ms", {k: round(v["avg_ms"], 3) for k, v in d["kernel_ms_in_schedule"].items() if "Sync" in k or "Resample" in k})
PY
done
