"""Points dali_amd._capi at the hipemu build (tools/hipemu): the kernel sources of dali_amd/csrc compiled as plain C++
against a CPU model of the HIP constructs they use.  TEST INFRASTRUCTURE: imported by tests only; the product never
looks for these libraries and fails loudly without the gfx950 build, as before."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tools", "hipemu")


def build(san=""):
    """san: "" (plain), "address" / "undefined" (sanitizer builds of the kernels; LD_PRELOAD the clang runtime), "race"
    (the kernels' memory accesses go through the lane-level race detector, hipemuRaceCount())."""
    extra = ["RACE=1"] if san == "race" else ([f"SAN={san}"] if san else [])
    # a build of the kernels with other -DDALIAMD_... knobs (tools/hipemu/ldsprof.py compares LDS layouts with it)
    tag, flags = os.environ.get("HIPEMU_TAG", ""), os.environ.get("HIPEMU_EXTRA", "")
    if tag:
        extra += [f"TAG={tag}", f"EXTRA={flags}"]
    subprocess.check_call(["make", "-s", "-j8", "-C", EMU_DIR] + extra)
    return os.path.join(EMU_DIR, "_build" + (f"_{san}" if san else "") + (f"_{tag}" if tag else ""), "lib")


def activate(san=""):
    """Must run before the first dali_amd._capi.lib() / host() call of the process."""
    libdir = build(san)
    from dali_amd import _capi
    if getattr(_capi, "_kernels", None) is not None or getattr(_capi, "_host", None) is not None:
        raise RuntimeError("dali_amd._capi has already loaded the product libraries in this process")
    _capi.KERNELS_LIB = os.path.join(libdir, "libdali_amd_kernels.so")
    _capi.HOST_LIB = os.path.join(libdir, "libdali_amd_host.so")
    patch_torch()
    return libdir


class _FakeStream:
    cuda_stream = 0

    def __init__(self, *a, **k):
        pass

    def synchronize(self):
        pass

    def wait_stream(self, other):
        pass

    def wait_event(self, ev):
        pass

    def record_event(self, ev=None):
        return ev or _FakeEvent()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _FakeEvent:
    def __init__(self, *a, **k):
        import time
        self._t = time.perf_counter()

    def record(self, stream=None):
        import time
        self._t = time.perf_counter()

    def synchronize(self):
        pass

    def wait(self, stream=None):
        pass

    def query(self):
        return True

    def elapsed_time(self, other):
        return (other._t - self._t) * 1e3


def patch_torch():
    """"Device" memory of the CPU model is host memory: torch's cuda factories and stream calls become their CPU
    equivalents (tests only)."""
    import ctypes as C
    import contextlib
    import numpy as np
    import torch

    if getattr(torch, "_hipemu_patched", False):
        return

    def is_cuda_dev(d):
        try:
            return d is not None and torch.device(d).type == "cuda"
        except (RuntimeError, TypeError):
            return False

    def wrap_factory(orig):
        def f(*a, **k):
            if is_cuda_dev(k.get("device")):
                k["device"] = "cpu"
            k.pop("pin_memory", None)
            return orig(*a, **k)
        return f

    for name in ("zeros", "empty", "ones", "full", "tensor", "arange", "randint", "rand", "randn", "empty_like",
                 "zeros_like", "ones_like", "full_like", "empty_strided"):
        setattr(torch, name, wrap_factory(getattr(torch, name)))

    orig_as_tensor = torch.as_tensor

    def as_tensor(obj, *a, **k):
        iface = getattr(obj, "__cuda_array_interface__", None)
        if is_cuda_dev(k.get("device")):
            k["device"] = "cpu"
        if iface is not None:
            np_t = np.dtype(iface["typestr"])
            shape = tuple(int(v) for v in iface["shape"])
            strides = iface.get("strides")
            n = int(np.prod(shape)) if shape else 1
            if n == 0:
                return torch.empty(shape, dtype=torch.from_numpy(np.zeros(0, np_t)).dtype)
            if strides is None:
                extent = n * np_t.itemsize
            else:
                extent = sum((s - 1) * abs(int(st)) for s, st in zip(shape, strides)) + np_t.itemsize
            buf = (C.c_char * extent).from_address(int(iface["data"][0]))
            arr = np.ndarray(shape, np_t, buffer=buf, strides=None if strides is None else tuple(int(s) for s in strides))
            t = torch.from_numpy(arr)
            t._hipemu_owner = obj
            return t
        return orig_as_tensor(obj, *a, **k)

    torch.as_tensor = as_tensor
    torch.Tensor.cuda = lambda self, *a, **k: self.clone()
    torch.Tensor.cpu = lambda self, *a, **k: self.clone()   # a device -> host transfer is a copy
    orig_to = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, (str, torch.device)) and is_cuda_dev(x)) else x for x in a)
        if is_cuda_dev(k.get("device")):
            k["device"] = "cpu"
        return orig_to(self, *a, **k)

    torch.Tensor.to = to
    torch.Tensor.record_stream = lambda self, s: None
    torch.Tensor.pin_memory = lambda self, *a, **k: self
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.current_stream = lambda *a, **k: _FakeStream()
    torch.cuda.default_stream = lambda *a, **k: _FakeStream()
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.cuda.Stream = _FakeStream
    torch.cuda.Event = _FakeEvent
    torch.cuda.is_available = lambda: True
    torch.cuda.device_count = lambda: 1
    torch.cuda.set_device = lambda d: None
    torch.cuda.current_device = lambda: 0
    torch._hipemu_patched = True
