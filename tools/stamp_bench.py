"""Development probe: runs bench.py's main() in this process with a -DDALIAMD_EXP_STAMPS kernel library and dumps the
stamps of the LAST SyncKernel / IndexedSyncKernel launch (tools/stamp_probe.py explains the columns).
    python tools/stamp_bench.py OUTDIR <bench.py arguments>"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = sys.argv[1]
sys.argv = ["bench.py"] + sys.argv[2:]
import bench  # noqa: E402

try:
    bench.main()
finally:
    from dali_amd import _capi
    lib = C.CDLL(_capi.KERNELS_LIB)
    lib.daliamdDebugReadStamps.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_int]
    os.makedirs(out, exist_ok=True)
    for which, name in ((0, "sync"), (1, "indexed"), (2, "block")):
        a = np.zeros((8192, 16), np.uint64)
        assert lib.daliamdDebugReadStamps(which, a.ctypes.data, a.nbytes, 0) == 0
        a = a[a[:, 0] != 0]
        if len(a):   # the rows of the LAST launch only (earlier launches with more workgroups leave stale rows behind)
            a = a[a[:, 0].astype(np.int64) > np.int64(a[:, 0].max()) - 50000]
        np.save(os.path.join(out, f"stamps_{name}.npy"), a)
        if len(a) == 0:
            continue
        t0 = a[:, 0].min()
        cols = [c for c in (range(1, 13) if name != "block" else (1, 2)) if (a[:, c] != 0).any()]
        print(f"{name}: {len(a)} workgroups, span {(max(a[:, c].max() for c in cols) - t0) / 100:.1f} us", file=sys.stderr)
        for c in cols:
            ok = a[:, c] != 0
            d = (a[ok, c].astype(np.int64) - a[ok, 0].astype(np.int64)) / 100.0
            print("  col %2d: n %4d  since wg start p50 %.1f p90 %.1f max %.1f us; absolute max %.1f" %
                  (c, ok.sum(), *np.percentile(d, [50, 90, 100]), (a[ok, c].max() - t0) / 100.0), file=sys.stderr)
        st = (a[:, 0].astype(np.int64) - np.int64(t0)) / 100.0
        print("  start p50 %.1f p90 %.1f max %.1f us" % tuple(np.percentile(st, [50, 90, 100])), file=sys.stderr)
        key = (a[:, 13] & 0xF) * 1000 + ((a[:, 14] >> 13) & 7) * 100 + ((a[:, 14] >> 12) & 1) * 50 + ((a[:, 14] >> 8) & 0xF)
        u, n = np.unique(key, return_counts=True)
        print(f"  CUs {len(u)}, workgroups per CU hist {np.bincount(n).tolist()}", file=sys.stderr)
        if name == "block":
            # concurrency over time: workgroups alive per 5-us bin; tasks per workgroup
            end = a[:, 2].astype(np.int64)
            bins = np.arange(0, (end.max() - t0) / 100.0 + 5, 5)
            alive = [int(((st <= b) & ((end - np.int64(t0)) / 100.0 > b)).sum()) for b in bins]
            print("  alive per 5 us:", alive, file=sys.stderr)
            dur = (end - a[:, 0].astype(np.int64)) / 100.0
            tasks = a[:, 12] // 1000 + a[:, 12] % 1000
            print("  duration p10 %.1f p50 %.1f p90 %.1f max %.1f us; tasks/wg p50 %d max %d; us per task p50 %.2f" %
                  (*np.percentile(dur, [10, 50, 90, 100]), np.median(tasks), tasks.max(), np.median(dur / np.maximum(tasks, 1))), file=sys.stderr)
