"""Feasibility probe (GPU box): can the page-cache pages of a read-only file mapping be registered with HIP and read by
the device without a staging copy?  hipHostRegister on a MAP_SHARED / PROT_READ mapping, then (a) one large async H2D copy,
(b) 256 copies of 94 KB, (c) a device-side copy (hipMemcpyDtoD from the mapping's device pointer), each against the same
from torch pinned memory."""
import ctypes as C
import mmap
import os
import time

import numpy as np
import torch

hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipHostGetDevicePointer.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipGetErrorString.restype = C.c_char_p
hip.hipGetErrorString.argtypes = [C.c_int]

N = 64 << 20
path = "/tmp/hostreg_probe.bin"
data = np.random.default_rng(1).integers(0, 256, N, dtype=np.uint8)
data.tofile(path)
dev = torch.empty(N, dtype=torch.uint8, device="cuda")
pinned = torch.from_numpy(data.copy()).pin_memory()
torch.cuda.synchronize()
stream = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def copies(src_ptr, chunk):
    def run():
        for off in range(0, N, chunk):
            n = min(chunk, N - off)
            rc = hip.hipMemcpyAsync(dev.data_ptr() + off, src_ptr + off, n, 1, stream)
            assert rc == 0, hip.hipGetErrorString(rc)
    return run


print("pinned: one copy %.2f GB/s; 94 KB copies %.2f GB/s (%.1f us each)" % (
    N / timed(copies(pinned.data_ptr(), N)) / 1e9, N / timed(copies(pinned.data_ptr(), 94 << 10)) / 1e9,
    timed(copies(pinned.data_ptr(), 94 << 10)) / (N / (94 << 10)) * 1e6))

for prot, name in ((mmap.PROT_READ, "PROT_READ"), (mmap.PROT_READ | mmap.PROT_WRITE, "PROT_READ|WRITE (MAP_PRIVATE)")):
    fd = os.open(path, os.O_RDONLY)
    flags = mmap.MAP_SHARED if prot == mmap.PROT_READ else mmap.MAP_PRIVATE
    m = mmap.mmap(fd, N, flags=flags, prot=prot)
    arr = np.frombuffer(m, np.uint8)
    _ = int(arr[::4096].sum())                       # fault the pages in
    ptr = arr.ctypes.data
    for fl, fname in ((0, "default"), (8, "read-only"), (2, "mapped"), (10, "mapped|read-only")):
        t0 = time.perf_counter()
        rc = hip.hipHostRegister(ptr, N, fl)
        t_reg = time.perf_counter() - t0
        if rc != 0:
            print(f"{name}: hipHostRegister(flags={fname}) FAILED: {hip.hipGetErrorString(rc).decode()}")
            continue
        try:
            dev.zero_()
            one = timed(copies(ptr, N))
            ok = bool(torch.equal(dev.cpu(), torch.from_numpy(data)))
            small = timed(copies(ptr, 94 << 10))
            dp = C.c_void_p()
            rc2 = hip.hipHostGetDevicePointer(C.byref(dp), ptr, 0)
            d2d = None
            if rc2 == 0 and dp.value:
                def dd():
                    rc = hip.hipMemcpyAsync(dev.data_ptr(), dp.value, N, 3, stream)
                    assert rc == 0, hip.hipGetErrorString(rc)
                dev.zero_()
                d2d = timed(dd)
                ok = ok and bool(torch.equal(dev.cpu(), torch.from_numpy(data)))
            print(f"{name}: register({fname}) ok in {t_reg * 1e3:.1f} ms ({t_reg / (N >> 20) * 1e6:.0f} us/MiB); one copy {N / one / 1e9:.2f} GB/s; "
                  f"94 KB copies {N / small / 1e9:.2f} GB/s; device-side copy "
                  f"{'n/a (rc %d)' % rc2 if d2d is None else '%.2f GB/s' % (N / d2d / 1e9)}; data {'equal' if ok else 'DIFFER'}")
        finally:
            hip.hipHostUnregister(ptr)
    del arr
    m.close()
    os.close(fd)
os.unlink(path)
