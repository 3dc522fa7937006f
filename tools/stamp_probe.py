"""Development probe (GPU box): per-workgroup wall-clock stamps of SyncKernel / IndexedSyncKernel on the bench's first
batch, from a kernel library built with -DDALIAMD_EXP_STAMPS (tools/build_variant.sh stamps jpeg_huffman.hip
"-DDALIAMD_EXP_STAMPS=1", copied over dali_amd/lib/libdali_amd_kernels.so by the calling script).
    python tools/stamp_probe.py OUTDIR
Writes OUTDIR/stamps_sync.npy, stamps_indexed.npy ([workgroups, 16] u64, 100 MHz ticks) and prints a summary."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dali_amd.testing import synth_dataset  # noqa: E402  (forks: before torch)

enc = synth_dataset(0, 256, seed=1234, workers=8)
import torch  # noqa: E402
from dali_amd import _capi, backend as B  # noqa: E402

lib = C.CDLL(_capi.KERNELS_LIB)
lib.daliamdDebugReadStamps.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_int]
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stamps"
os.makedirs(out, exist_ok=True)


def read(which):
    a = np.zeros((8192, 16), np.uint64)
    rc = lib.daliamdDebugReadStamps(which, a.ctypes.data, a.nbytes, 1)
    assert rc == 0, rc
    return a[a[:, 0] != 0]


def summary(name, a, cols):
    t0 = a[:, 0].min()
    end = max(a[:, c].max() for c in cols)
    print(f"{name}: {len(a)} workgroups, launch span {(end - t0) / 100:.1f} us")
    rel = lambda c: (a[:, c].astype(np.int64) - np.int64(t0)) / 100.0
    st = rel(0)
    print("  start      p50 %.1f p90 %.1f max %.1f us" % tuple(np.percentile(st, [50, 90, 100])))
    for c in cols:
        ok = a[:, c] != 0
        if not ok.any():
            continue
        d = (a[ok, c].astype(np.int64) - a[ok, 0].astype(np.int64)) / 100.0
        print("  col %2d: n %4d  since wg start p50 %.1f p90 %.1f max %.1f us; absolute max %.1f us" %
              (c, ok.sum(), *np.percentile(d, [50, 90, 100]), rel(c)[ok].max()))
    xcc = a[:, 13] & 0xF
    cu = (a[:, 14] >> 8) & 0xF
    se = (a[:, 14] >> 13) & 0x7
    sh = (a[:, 14] >> 12) & 0x1
    key = xcc * 1000 + se * 100 + sh * 50 + cu
    u, n = np.unique(key, return_counts=True)
    print(f"  distinct (xcc, se, sh, cu): {len(u)}; workgroups per CU: max {n.max()}, hist {np.bincount(n).tolist()}")


for rep in range(3):
    first, plan = B.decode_jpeg_batch(enc, device="cuda", exact_scan=False, index="build")
    torch.cuda.synchronize()
    s = read(0)
    read(1)
    again, _ = B.decode_jpeg_batch(enc, device="cuda", exact_scan=False, index="use", index_from=plan)
    torch.cuda.synchronize()
    i = read(1)
    read(0)
    if rep == 0:
        for a, b in zip(first, again):
            assert torch.equal(a, b)
np.save(os.path.join(out, "stamps_sync.npy"), s)
np.save(os.path.join(out, "stamps_indexed.npy"), i)
summary("SyncKernel", s, [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
summary("IndexedSyncKernel", i, [1, 2, 3, 4, 5])
