#!/bin/bash
# configs[3] evidence: the fused spectrogram -> mel -> dB launch, MFMA variant against the banded VALU variant and against
# the three separate kernels; kernel stats, HBM traffic and the matrix-core counters.
#   bash tools/audio_mfma_prof.sh r03        -> gpurun_out/<tag>_audio/*
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
OUT=$R/gpurun_out/${TAG}_audio
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in mfma valu unfused; do
  export DALI_AMD_MEL_VALU=0 DALI_AMD_NO_AUDIO_FUSION=0
  [ $V = valu ] && export DALI_AMD_MEL_VALU=1
  [ $V = unfused ] && export DALI_AMD_NO_AUDIO_FUSION=1
  python $R/bench.py --full-line --workload audio --steps 30 --warmup 5 > $OUT/bench_$V.json 2> $OUT/bench_$V.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$V -- python $R/bench.py --full-line --workload audio --steps 20 --warmup 3 > /dev/null 2> $OUT/stats_$V.log
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${V}_$C -- python $R/bench.py --full-line --workload audio --steps 5 --warmup 1 > /dev/null 2> $OUT/pmc_${V}_$C.log
  done
done
export DALI_AMD_MEL_VALU=0 DALI_AMD_NO_AUDIO_FUSION=0
# matrix-core counters of the MFMA variant (own pass; no trace domains next to --pmc)
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_mfma_SQ -- python $R/bench.py --full-line --workload audio --steps 5 --warmup 1 > /dev/null 2> $OUT/pmc_mfma_SQ.log
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_mfma_SQ2 -- python $R/bench.py --full-line --workload audio --steps 5 --warmup 1 > /dev/null 2> $OUT/pmc_mfma_SQ2.log
python - <<PY
import csv, glob, json, collections
out = {}
for v in ("mfma", "valu", "unfused"):
    res = collections.defaultdict(dict)
    for f in glob.glob("$OUT/stats_%s/**/*kernel_stats.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "daliamd" in r["Name"]:
                res[r["Name"].split("(")[0].replace("void ", "").replace("daliamd::", "")].update(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(list)
        for f in glob.glob("$OUT/pmc_%s_%s/**/*counter_collection.csv" % (v, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c and "daliamd" in r["Kernel_Name"]:
                    acc[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("daliamd::", "")].append(float(r["Counter_Value"]))
        for k, vals in acc.items():
            res[k][c + "_MB_per_launch"] = (2 if c == "FETCH_SIZE" else 1) * 1024 * sum(vals) / len(vals) / 1e6   # gfx950: FETCH_SIZE counts half
    out[v] = res
    try:
        out[v]["bench_line"] = json.loads(open("$OUT/bench_%s.json" % v).read().strip().splitlines()[-1])
    except Exception as e:
        out[v]["bench_line"] = str(e)
sq = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_mfma_SQ", "pmc_mfma_SQ2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "daliamd" in r["Kernel_Name"]:
                sq[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("daliamd::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out["mfma_counters"] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in sq.items()}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
for v in ("mfma", "valu", "unfused"):
    print(v, {k: {a: round(b, 1) for a, b in d.items() if not isinstance(b, (dict, str))} for k, d in out[v].items() if k != "bench_line"})
print("counters", out["mfma_counters"])
PY
