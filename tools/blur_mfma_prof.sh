#!/bin/bash
# configs[2] evidence for the matrix-core blur: kernel stats, HBM traffic, matrix-core and issue counters.
#   bash tools/blur_mfma_prof.sh r05        -> gpurun_out/<tag>_blur/summary.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05}
OUT=$R/gpurun_out/${TAG}_blur
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DALI_AMD_BLUR_MFMA=${DALI_AMD_BLUR_MFMA:-1}
CMD="python $R/bench.py --full-line --workload heavy_aug --steps 5 --warmup 1 --no-cpu-baseline --inflight 1"
pass() { timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$1 -- $CMD > /dev/null 2> $OUT/pmc_$1.log; }
pass SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32
pass SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
pass SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
pass SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
pass SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
pass FETCH_SIZE
pass WRITE_SIZE
python - <<PY
import csv, glob, json, collections
sq = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "daliamd" in r["Kernel_Name"]:
            sq[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("daliamd::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in sq.items()}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
for k, d in out.items():
    print(k, {c: round(v) for c, v in d.items()})
PY
