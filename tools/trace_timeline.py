"""Prints the overlap statistics and the dispatch timeline of the timed region of a trace written by tools/gpu_trace.sh.
   python tools/trace_timeline.py TRACE.csv.gz [first_batch [batches [lines]]]"""
import collections
import csv
import gzip
import sys

R = []
for q, d, n, s, e in csv.reader(gzip.open(sys.argv[1], "rt")):
    R.append(dict(q=q, d=d, n=n[:22], s=int(s), e=int(e)))
first = int(sys.argv[2]) if len(sys.argv) > 2 else -30
count = int(sys.argv[3]) if len(sys.argv) > 3 else 16
lines = int(sys.argv[4]) if len(sys.argv) > 4 else 120
sy = sorted([r for r in R if r["n"].startswith("SyncKernel")], key=lambda r: r["s"])
print(len(R), "dispatches,", len(sy), "batches")
t0, t1 = sy[first]["s"], sy[first + count]["s"]
reg = sorted([r for r in R if t0 <= r["s"] < t1], key=lambda r: r["s"])
print("region %.3f ms, %d dispatches, %.1f us per batch" % ((t1 - t0) / 1e6, len(reg), (t1 - t0) / 1e3 / count))
ev = []
for r in reg:
    ev.append((r["s"], 1, r))
    ev.append((r["e"], -1, r))
ev.sort(key=lambda x: (x[0], x[1]))
c, last, hist, active, solo = 0, t0, collections.Counter(), {}, collections.Counter()
for t, d, r in ev:
    hist[c] += t - last
    if c == 1:
        solo[next(iter(active.values()))] += t - last
    last = t
    c += d
    if d == 1:
        active[r["d"]] = r["n"]
    else:
        active.pop(r["d"], None)
tot = sum(hist.values())
print("kernels running at once:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
print("alone (us per batch):", {k: round(v / 1e3 / count) for k, v in solo.most_common()})
dur, cnt = collections.Counter(), collections.Counter()
for r in reg:
    dur[r["n"]] += r["e"] - r["s"]
    cnt[r["n"]] += 1
print("average duration (us):", {k: round(v / cnt[k] / 1e3, 1) for k, v in dur.most_common(12)})
for r in reg[:lines]:
    print(r["q"], "%9.1f %8.1f %9.1f" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, (r["e"] - t0) / 1e3), r["n"])
