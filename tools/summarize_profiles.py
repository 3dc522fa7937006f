#!/usr/bin/env python
"""Condenses a tools/collect_profiles.sh run into the files kept under profiles/:
   <tag>_kernel_stats.csv   rocprofv3 --stats table of the default bench command (all kernels)
   <tag>_traffic.json       per kernel: launches, average duration, FETCH_SIZE / WRITE_SIZE per launch
FETCH_SIZE/WRITE_SIZE are reported by rocprofv3 in KiB-like units of 1 KB... they are bytes/1024 on this ROCm; on
gfx950 FETCH_SIZE counts 128-byte read requests as 64 bytes (MI355X_MICROARCH.md, HBM section), so the corrected
read traffic is 2 x FETCH_SIZE; WRITE_SIZE is taken as reported (uncalibrated according to the same guide)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

out_dir, tag = sys.argv[1], sys.argv[2]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(out_dir if os.path.isabs(out_dir) else os.path.join(repo, out_dir))
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(repo, "gpurun_out", f"{tag}_summary")
os.makedirs(dst, exist_ok=True)


def short(name):
    return name.split("(")[0].replace("daliamd::", "").replace("void ", "")


stats = glob.glob(os.path.join(prof, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
if not stats:
    sys.exit(f"summarize_profiles: no *kernel_stats.csv under {prof}/stats - the rocprofv3 --stats pass failed (see stats.log)")
res = defaultdict(dict)
if stats:
    for r in csv.DictReader(open(stats[0])):
        res[short(r["Name"])].update(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), pct=float(r["Percentage"]))
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(prof, f"pmc_{counter}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    if not acc:
        sys.exit(f"summarize_profiles: the --pmc {counter} pass left no counter rows under {prof}/pmc_{counter} "
                 f"(see pmc_{counter}.log): refusing to write half a traffic file")
    for k, v in acc.items():
        res[k][f"{counter}_per_launch_KB"] = sum(v) / len(v)
for k, d in res.items():
    if "FETCH_SIZE_per_launch_KB" in d:
        d["read_bytes_per_launch_corrected"] = 2 * 1024 * d["FETCH_SIZE_per_launch_KB"]
    if "WRITE_SIZE_per_launch_KB" in d:
        d["write_bytes_per_launch"] = 1024 * d["WRITE_SIZE_per_launch_KB"]
    if "read_bytes_per_launch_corrected" in d and "write_bytes_per_launch" in d:
        d["hbm_bytes_per_launch"] = d["read_bytes_per_launch_corrected"] + d["write_bytes_per_launch"]
missing = [k for k, d in res.items() if "avg_ns" in d and d.get("pct", 0) >= 1.0 and "hbm_bytes_per_launch" not in d]
if missing:
    sys.exit(f"summarize_profiles: no FETCH_SIZE + WRITE_SIZE pair for {missing}")
regime = ("kernel_stats: the bench command as the driver runs it (several batches in flight, see --inflight: durations include the overlap "
          "with the other stream's kernels); FETCH_SIZE / WRITE_SIZE: separate passes with ONE batch in flight, so that a "
          "counter window holds exactly one kernel - bytes per launch do not depend on the overlap, durations do")
json.dump({"note": __doc__, "regime": regime, "kernels": res}, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
for k, d in sorted(res.items(), key=lambda kv: -kv[1].get("avg_ns", 0)):
    print(f"{k:40s} {d.get('avg_ns', 0) / 1e3:9.1f} us  hbm/launch {d.get('hbm_bytes_per_launch', float('nan')) / 1e6:10.1f} MB")
