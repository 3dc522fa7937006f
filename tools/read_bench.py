"""Development aid: cost of read() of a page-cached 94 KB file into pageable vs page-locked (hipHostMalloc) memory,
single thread and a 16-thread pool - what the file reader pays per sample."""
import ctypes as C
import os
import sys
import tempfile
import threading
import time

sys.path.insert(0, ".")
from dali_amd import _capi as capi  # noqa: E402

lib = capi.kernels()
libc = C.CDLL("libc.so.6", use_errno=True)
libc.read.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
libc.read.restype = C.c_ssize_t
size = 94 * 1024
d = tempfile.mkdtemp()
paths = []
for i in range(256):
    p = os.path.join(d, f"{i}.bin")
    open(p, "wb").write(os.urandom(size))
    paths.append(p.encode())
pinned = C.c_void_p()
assert lib.daliamdHostAlloc(C.byref(pinned), C.c_size_t(256 * size)) == 0
pageable = C.create_string_buffer(256 * size)


def read_all(base, lo, hi):
    for i in range(lo, hi):
        fd = os.open(paths[i], os.O_RDONLY)
        libc.read(fd, C.c_void_p(base + i * size), size)
        os.close(fd)


for name, base in (("pageable", C.addressof(pageable)), ("pinned", pinned.value)):
    for nthreads in (1, 16):
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            ths = [threading.Thread(target=read_all, args=(base, k * 256 // nthreads, (k + 1) * 256 // nthreads)) for k in range(nthreads)]
            [t.start() for t in ths]
            [t.join() for t in ths]
            best = min(best, time.perf_counter() - t0)
        print(f"{name:9s} threads {nthreads:2d}: {best * 1e3:.3f} ms per 256 files  ({256 * size / best / 1e9:.1f} GB/s)")
# memcpy within user space for reference
src = C.create_string_buffer(256 * size)
for name, base in (("pageable", C.addressof(pageable)), ("pinned", pinned.value)):
    t0 = time.perf_counter()
    for _ in range(10):
        C.memmove(base, src, 256 * size)
    dt = (time.perf_counter() - t0) / 10
    print(f"memcpy to {name}: {dt * 1e3:.3f} ms ({256 * size / dt / 1e9:.1f} GB/s)")
