"""GPU: the batched device-side copy (daliamdGatherCopy) and host memory registered for in-place device reads
(daliamdHostRegister) - what the mixed decoders fetch the encoded files of a batch with when the reader hands out its
file mappings instead of copies (round 5).  Bit-exact: every byte of every record, nothing outside."""
import ctypes as C
import mmap
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gather(records, table_device=True):
    """records: (src address, dst address, bytes); runs one launch on the current stream."""
    from dali_amd import _capi as capi
    lib = capi.kernels()
    n = len(records)
    arr = (capi.GatherDesc * max(n, 1))()
    for i, (s, d, b) in enumerate(records):
        arr[i].src, arr[i].dst, arr[i].bytes, arr[i].reserved = s, d, b, 0
    raw = np.frombuffer(arr, np.uint8).copy()
    tab = torch.from_numpy(raw).cuda() if table_device else torch.from_numpy(raw).pin_memory()
    lib.daliamdGatherCopy.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    capi.check(lib.daliamdGatherCopy(tab.data_ptr(), n, max([b for _, _, b in records] + [0]),
                                     torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return tab


@pytest.mark.parametrize("src_kind", ["device", "pinned"])
def test_records_of_every_length_and_alignment(src_kind):
    rng = np.random.default_rng(5)
    data = torch.from_numpy(rng.integers(0, 256, 1 << 20, dtype=np.uint8))
    src = data.cuda() if src_kind == "device" else data.pin_memory()
    dst = torch.full((3 << 20,), 0xA5, dtype=torch.uint8, device="cuda")
    lens = [0, 1, 2, 15, 16, 17, 31, 32, 33, 255, 4095, 4096, 4097, 16383, 16384, 16385, 16400, 40000, 94652, 200001]
    recs, want, o = [], [], 64
    for k, n in enumerate(lens):
        for sa, da in ((0, 0), (1, 1), (5, 0), (0, 7), (9, 3), (15, 15)):
            so = int(rng.integers(0, (1 << 20) - n - 32)) // 16 * 16 + sa
            do = (o + 15) // 16 * 16 + da
            recs.append((src.data_ptr() + so, dst.data_ptr() + do, n))
            want.append((so, do, n))
            o = do + n + 48
    assert o < dst.numel()
    _gather(recs, table_device=(src_kind == "device"))
    got = dst.cpu().numpy()
    ref = np.full(dst.numel(), 0xA5, np.uint8)
    d = data.numpy()
    for so, do, n in want:
        ref[do:do + n] = d[so:so + n]
    assert np.array_equal(got, ref), np.flatnonzero(got != ref)[:8]


def test_empty_table_and_empty_records():
    from dali_amd import _capi as capi
    lib = capi.kernels()
    lib.daliamdGatherCopy.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    capi.check(lib.daliamdGatherCopy(None, 0, 0, torch.cuda.current_stream().cuda_stream))
    dst = torch.zeros(64, dtype=torch.uint8, device="cuda")
    src = torch.ones(64, dtype=torch.uint8, device="cuda")
    _gather([(src.data_ptr(), dst.data_ptr(), 0)] * 3)
    assert int(dst.sum()) == 0


def test_registered_file_mapping_is_read_in_place(tmp_path):
    """A read-only mapping of a file, registered: the device copies straight out of the page cache - to the file's very
    last byte and not one further (the last page ends where the mapping ends)."""
    from dali_amd import _capi as capi
    lib = capi.kernels()
    lib.daliamdHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
    lib.daliamdHostUnregister.argtypes = [C.c_void_p]
    rng = np.random.default_rng(6)
    for size in (3 * 4096, 94652, 1):             # a whole number of pages; an ordinary file; one byte
        blob = rng.integers(0, 256, size, dtype=np.uint8)
        p = tmp_path / f"f{size}.bin"
        p.write_bytes(blob.tobytes())
        fd = os.open(p, os.O_RDONLY)
        m = mmap.mmap(fd, size, flags=mmap.MAP_SHARED, prot=mmap.PROT_READ)
        view = np.frombuffer(m, np.uint8)
        ptr = view.ctypes.data
        same = C.c_int(0)
        capi.check(lib.daliamdHostRegister(ptr, size, C.byref(same)))
        try:
            assert same.value == 1, "the device does not address registered host memory at its host address"
            dst = torch.zeros(size + 32, dtype=torch.uint8, device="cuda")
            off = min(623, size - 1)               # (an entropy-coded segment starts somewhere behind the headers)
            _gather([(ptr + off, dst.data_ptr() + 16 + (off & 15), size - off)])
            got = dst.cpu().numpy()
            lo = 16 + (off & 15)
            assert np.array_equal(got[lo:lo + size - off], blob[off:])
            assert not got[:lo].any() and not got[lo + size - off:].any()
        finally:
            capi.check(lib.daliamdHostUnregister(ptr))
            del view
            m.close()
            os.close(fd)
