"""GPU box: daliamdGatherCopy out of registered file mappings when every launch reads OTHER files (2048 files, batches of
256 in turn) - against the same launches over one and the same batch: is the address translation of fresh page-cache pages
what a pipeline that walks through a data set pays for?"""
import ctypes as C
import mmap
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dali_amd import _capi as capi

lib = capi.kernels()
lib.daliamdHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
lib.daliamdGatherCopy.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
rng = np.random.default_rng(3)
F, N, SZ = 2048, 256, 94652
d = "/tmp/gather_probe2"
os.makedirs(d, exist_ok=True)
blob = rng.integers(0, 256, SZ, dtype=np.uint8)
keep, ptrs = [], []
for i in range(F):
    p = os.path.join(d, f"{i}.bin")
    blob.tofile(p)
    fd = os.open(p, os.O_RDONLY)
    m = mmap.mmap(fd, SZ, flags=mmap.MAP_SHARED | getattr(mmap, "MAP_POPULATE", 0), prot=mmap.PROT_READ)
    v = np.frombuffer(m, np.uint8)
    same = C.c_int(0)
    capi.check(lib.daliamdHostRegister(v.ctypes.data, SZ, C.byref(same)))
    keep.append((m, v, fd))
    ptrs.append(v.ctypes.data)
dst = torch.empty(N * (SZ + 64), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
tabs = []
for b in range(F // N):
    arr = (capi.GatherDesc * N)()
    for i in range(N):
        s = ptrs[b * N + i] + 623
        arr[i].src, arr[i].dst, arr[i].bytes = s, dst.data_ptr() + i * (SZ + 64) + 16 + (s & 15), SZ - 623
    tabs.append(torch.from_numpy(np.frombuffer(arr, np.uint8).copy()).cuda())
total = N * (SZ - 623)


def rate(seq):
    for t in seq[:2]:
        capi.check(lib.daliamdGatherCopy(t.data_ptr(), N, SZ, st))
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for t in seq:
        capi.check(lib.daliamdGatherCopy(t.data_ptr(), N, SZ, st))
    b.record()
    torch.cuda.synchronize()
    return total * len(seq) / (a.elapsed_time(b) * 1e-3) / 1e9


print("the same 256 files every launch: %.1f GB/s" % rate([tabs[0]] * 24))
print("other files every launch (2048): %.1f GB/s" % rate(tabs * 3))
print("the same 256 files again:        %.1f GB/s" % rate([tabs[3]] * 24))
