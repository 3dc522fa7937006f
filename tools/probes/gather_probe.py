"""GPU box: rate of daliamdGatherCopy out of 256 registered file mappings (94 KB each, page-cache pages) against one
device-side hipMemcpy of the same bytes out of one registered mapping, and out of pinned memory."""
import ctypes as C
import mmap
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dali_amd import _capi as capi

lib = capi.kernels()
lib.daliamdHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
lib.daliamdGatherCopy.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
rng = np.random.default_rng(3)
N, SZ = 256, 94652
d = "/tmp/gather_probe"
os.makedirs(d, exist_ok=True)
maps, recs = [], []
dst = torch.empty(N * (SZ + 64), dtype=torch.uint8, device="cuda")
t_reg = 0.0
for i in range(N):
    p = os.path.join(d, f"{i}.bin")
    rng.integers(0, 256, SZ, dtype=np.uint8).tofile(p)
    fd = os.open(p, os.O_RDONLY)
    m = mmap.mmap(fd, SZ, flags=mmap.MAP_SHARED | getattr(mmap, "MAP_POPULATE", 0), prot=mmap.PROT_READ)
    v = np.frombuffer(m, np.uint8)
    same = C.c_int(0)
    t0 = time.perf_counter()
    capi.check(lib.daliamdHostRegister(v.ctypes.data, SZ, C.byref(same)))
    t_reg += time.perf_counter() - t0
    assert same.value == 1
    maps.append((m, v, fd))
    off = 623
    recs.append((v.ctypes.data + off, dst.data_ptr() + i * (SZ + 64) + 16 + ((v.ctypes.data + off) & 15), SZ - off))
print("registration: %.1f us per 94 KB file" % (t_reg / N * 1e6))
arr = (capi.GatherDesc * N)()
for i, (s, t, b) in enumerate(recs):
    arr[i].src, arr[i].dst, arr[i].bytes = s, t, b
tab = torch.from_numpy(np.frombuffer(arr, np.uint8).copy()).cuda()
st = torch.cuda.current_stream().cuda_stream
total = sum(b for _, _, b in recs)


def rate(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return total * reps / (a.elapsed_time(b) * 1e-3) / 1e9


print("gather out of 256 registered mappings: %.1f GB/s" % rate(lambda: capi.check(lib.daliamdGatherCopy(tab.data_ptr(), N, SZ, st))))
ok = all(np.array_equal(dst[t - dst.data_ptr():t - dst.data_ptr() + b].cpu().numpy(), maps[i][1][623:]) for i, (s, t, b) in enumerate(recs[:8]))
print("data", "equal" if ok else "DIFFER")
# the same kernel out of pinned memory (one block)
pin = torch.from_numpy(rng.integers(0, 256, N * SZ, dtype=np.uint8)).pin_memory()
for i in range(N):
    arr[i].src = pin.data_ptr() + i * SZ + 623
tab2 = torch.from_numpy(np.frombuffer(arr, np.uint8).copy()).cuda()
print("gather out of one pinned block:       %.1f GB/s" % rate(lambda: capi.check(lib.daliamdGatherCopy(tab2.data_ptr(), N, SZ, st))))
devsrc = pin.cuda()
for i in range(N):
    arr[i].src = devsrc.data_ptr() + i * SZ + 623
tab3 = torch.from_numpy(np.frombuffer(arr, np.uint8).copy()).cuda()
print("gather out of device memory:          %.1f GB/s" % rate(lambda: capi.check(lib.daliamdGatherCopy(tab3.data_ptr(), N, SZ, st))))
d2 = torch.empty(N * SZ, dtype=torch.uint8, device="cuda")
print("copy engine, pinned block -> device:  %.1f GB/s" % (rate(lambda: d2.copy_(pin, non_blocking=True)) * (N * SZ) / total))
