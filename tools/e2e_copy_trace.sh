#!/bin/bash
# GPU box: rocprofv3 kernel + memory-copy trace of the end-to-end leg alone (tools/e2e_only.py): how long a batch's host->device
# transfer lasts, how long the copy engine idles between two of them, what runs meanwhile.   bash tools/e2e_copy_trace.sh TAG [ENV...]
TAG=${1:-e2e_trace}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --marker-trace --output-format csv -d /tmp/e2e_tr -- python $R/tools/e2e_only.py > $OUT/run.log 2> $OUT/run.err
tail -3 $OUT/run.log
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
cp = glob.glob("/tmp/e2e_tr/**/*memory_copy_trace.csv", recursive=True)
kt = glob.glob("/tmp/e2e_tr/**/*kernel_trace.csv", recursive=True)
print("files", cp, kt)
rows = list(csv.DictReader(open(cp[0])))
print("columns", list(rows[0].keys()))
print("first rows", rows[:3])
big = [r for r in rows if "HOST_TO_DEVICE" in r.get("Direction", "").upper() or "H2D" in r.get("Direction", "").upper()]
def g(r, *names):
    for n in names:
        if n in r: return r[n]
st = lambda r: int(g(r, "Start_Timestamp", "start_timestamp")); en = lambda r: int(g(r, "End_Timestamp", "end_timestamp"))
sz = lambda r: 25.4e6   # (the trace has no size column: a batch's transfer is the 25 MB reader block)
big = sorted([r for r in big if en(r) - st(r) > 150e3], key=st)
print(len(rows), "copies,", len(big), "H2D longer than 150 us")
tail = big[-150:]
dur = [(en(r) - st(r)) / 1e3 for r in tail]; gap = [(st(b) - en(a)) / 1e3 for a, b in zip(tail, tail[1:])]; per = [(st(b) - st(a)) / 1e3 for a, b in zip(tail, tail[1:])]
import statistics as S
print("last 150 large H2D: MB %.1f  duration us mean %.0f p50 %.0f p90 %.0f | gap us mean %.0f p50 %.0f p90 %.0f | period us mean %.0f" % (
    S.mean(map(sz, tail)) / 1e6, S.mean(dur), S.median(dur), sorted(dur)[int(.9 * len(dur))], S.mean(gap), S.median(gap), sorted(gap)[int(.9 * len(gap))], S.mean(per)))
print("GB/s inside a copy %.1f, overall %.1f" % (S.mean(map(sz, tail)) / S.mean(dur) / 1e3, S.mean(map(sz, tail)) / S.mean(per) / 1e3))
# what ended just before each transfer started: a kernel the copy stream waited for (small, constant distance), or nothing
krows = list(csv.DictReader(open(kt[0])))
kname = lambda r: g(r, "Kernel_Name", "kernel_name")[:28]
kev = sorted(((int(g(r, "End_Timestamp")), kname(r)) for r in krows))
import bisect
ends = [e for e, _ in kev]
near = collections.Counter()
dist = collections.defaultdict(list)
for r in tail:
    i = bisect.bisect_right(ends, st(r)) - 1
    if i >= 0:
        near[kev[i][1]] += 1
        dist[kev[i][1]].append((st(r) - kev[i][0]) / 1e3)
print("last kernel to END before a transfer STARTS:", {k: (c, round(S.median(dist[k]), 1)) for k, c in near.most_common(6)}, "(count, median us before)")
# ... and what the previous transfer's END coincides with on the host side cannot be seen here; kernels that START right after a transfer ends:
kst = sorted(((int(g(r, "Start_Timestamp")), kname(r)) for r in krows))
starts = [e for e, _ in kst]
nxt = collections.Counter(); nd = collections.defaultdict(list)
for r in tail:
    i = bisect.bisect_left(starts, en(r))
    if i < len(kst):
        nxt[kst[i][1]] += 1; nd[kst[i][1]].append((kst[i][0] - en(r)) / 1e3)
print("first kernel to START after a transfer ENDS:", {k: (c, round(S.median(nd[k]), 1)) for k, c in nxt.most_common(6)})
# host side: roctx ranges of the stage threads (daliamdRangePush): which range was open when each idle gap of the copy engine began
mk = glob.glob("/tmp/e2e_tr/**/*marker_api_trace.csv", recursive=True)
if mk:
    mrows = list(csv.DictReader(open(mk[0])))
    print("marker columns", list(mrows[0].keys()), len(mrows))
    rng = [(int(g(r, "Start_Timestamp")), int(g(r, "End_Timestamp")), g(r, "Function", "Name", "Message") or "", g(r, "Thread_Id", "Tid")) for r in mrows]
    rng = [x for x in rng if x[1] > x[0]]
    names = collections.Counter(x[2] for x in rng)
    print("ranges:", names.most_common(12))
    t_lo, t_hi = st(tail[0]), en(tail[-1])
    for nm in [n for n, _ in names.most_common(12)]:
        d = [(b - a) / 1e3 for a, b, n, _ in rng if n == nm and t_lo <= a <= t_hi]
        if d:
            print("  %-44s n %4d  mean %7.1f us  p90 %7.1f" % (nm[:44], len(d), S.mean(d), sorted(d)[int(.9 * len(d))]))
    # per gap: the ranges open at the start of the gap
    opened = collections.Counter()
    for a, b in zip(tail, tail[1:]):
        t = en(a) + 1000
        for x in rng:
            if x[0] <= t < x[1]:
                opened[x[2][:44]] += 1
    print("open when the copy engine fell idle:", opened.most_common(10))
open(out + "/copies_tail.txt", "w").write("\n".join(f"{st(r)} {en(r)} {sz(r)}" for r in tail))
PY
