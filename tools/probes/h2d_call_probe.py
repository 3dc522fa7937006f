"""GPU box: does an asynchronous host->device transfer of a reader block (25 MB, page-locked) return at once, and do transfers
queued behind each other run back to back?  Host time of every call + the device time of the train."""
import ctypes as C
import time

import torch

from dali_amd import _capi as capi

lib = capi.kernels()
MB = 25
src = torch.empty(MB << 20, dtype=torch.uint8).pin_memory()
dst = [torch.empty(MB << 20, dtype=torch.uint8, device="cuda") for _ in range(4)]
s = C.c_void_p()
capi.check(lib.daliamdStreamCreate(C.byref(s), 1))
for _ in range(3):
    capi.check(lib.daliamdMemcpyH2DAsync(C.c_void_p(dst[0].data_ptr()), C.c_void_p(src.data_ptr()), C.c_size_t(MB << 20), s))
capi.check(lib.daliamdStreamSynchronize(s))
for spacing_ms in (0.0, 0.3, 0.45, 0.6):
    calls = []
    t0 = time.perf_counter()
    for k in range(20):
        a = time.perf_counter()
        capi.check(lib.daliamdMemcpyH2DAsync(C.c_void_p(dst[k % 4].data_ptr()), C.c_void_p(src.data_ptr()), C.c_size_t(MB << 20), s))
        calls.append(time.perf_counter() - a)
        while time.perf_counter() - a < spacing_ms * 1e-3:
            pass
    t_enq = time.perf_counter() - t0
    capi.check(lib.daliamdStreamSynchronize(s))
    el = time.perf_counter() - t0
    print(f"calls {spacing_ms:.2f} ms apart: host time per call median {1e6 * sorted(calls)[10]:.0f} us max {1e6 * max(calls):.0f} us; "
          f"enqueue loop {1e3 * t_enq:.2f} ms, all 20 done after {1e3 * el:.2f} ms = {20 * MB * 1.048576 / el / 1e3:.1f} GB/s")
