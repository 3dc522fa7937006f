"""Indexed JPEG containers (".didx": tools/jpeg2idx.py, dali_amd/host/jpeg_indexed.cpp, readers.file(index_path=...)) - the CPU
half: building and parsing the container, what the tool does with files the GPU decoder does not take, and the reader handing
out containers in place of files under the files' names.  The decode from a container is tests/test_gpu_jpeg_indexed_files.py;
the index entry itself is held byte for byte to the device-built one in tests/test_gpu_jpeg_index.py."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from dali_amd import _capi as capi
from tests.util import encode_jpeg, synth_image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _build(enc):
    host = capi.host()
    data = np.frombuffer(enc, np.uint8)
    n = C.c_size_t(0)
    rc = host.daliamdJpegIndexedBuild(data.ctypes.data_as(C.c_void_p), C.c_size_t(data.size), None, C.c_size_t(0), C.byref(n))
    if rc:
        return None, host.daliamdHostGetLastErrorMessage().decode()
    out = np.zeros(n.value, np.uint8)
    if host.daliamdJpegIndexedBuild(data.ctypes.data_as(C.c_void_p), C.c_size_t(data.size), out.ctypes.data_as(C.c_void_p),
                                    C.c_size_t(out.size), C.byref(n)):
        return None, host.daliamdHostGetLastErrorMessage().decode()      # (found while the entry is built: a truncated stream)
    return out, ""


class View(C.Structure):
    _fields_ = [("header", C.c_void_p), ("header_len", C.c_int32), ("ecs_len", C.c_int32), ("index_offset", C.c_int64),
                ("index_bytes", C.c_int64), ("jpeg_size", C.c_int64)]


@pytest.mark.parametrize("kw", [dict(subsampling="4:2:0"), dict(subsampling="4:4:4", optimize=True), dict(subsampling="4:2:2"), "gray"])
def test_container_holds_the_headers_and_an_index_entry(kw):
    rng = np.random.default_rng(3)
    enc = encode_jpeg(synth_image(rng, 120, 161, 1), 80) if kw == "gray" else encode_jpeg(synth_image(rng, 120, 161), 85, **kw)
    box, why = _build(enc)
    assert box is not None, why
    host = capi.host()
    assert host.daliamdJpegIndexedIs(box.ctypes.data_as(C.c_void_p), C.c_size_t(box.size)) == 1
    assert host.daliamdJpegIndexedIs(np.frombuffer(enc, np.uint8).ctypes.data_as(C.c_void_p), C.c_size_t(len(enc))) == 0
    v = View()
    capi.check_host(host.daliamdJpegIndexedParse(box.ctypes.data_as(C.c_void_p), C.c_size_t(box.size), C.byref(v)))
    assert v.jpeg_size == len(enc) and v.index_offset % 64 == 0 and v.index_offset + v.index_bytes == box.size
    # the headers are the file's, and the product's header analysis reads them like the file's
    assert bytes(box[64:64 + v.header_len]) == enc[:v.header_len]
    data = np.frombuffer(enc, np.uint8)
    i0, s0, i1, s1 = capi.JpegInfo(), capi.JpegScan(), capi.JpegInfo(), capi.JpegScan()
    capi.check_host(host.daliamdJpegParse(data.ctypes.data_as(C.c_void_p), C.c_size_t(data.size), C.byref(i0)))
    capi.check_host(host.daliamdJpegAnalyzeScan(data.ctypes.data_as(C.c_void_p), C.c_size_t(data.size), C.byref(i0), C.byref(s0)))
    capi.check_host(host.daliamdJpegAnalyzeHeader(C.c_void_p(v.header), C.c_size_t(v.header_len), C.byref(i1), C.byref(s1)))
    assert bytes(i0) == bytes(i1) and s1.eligible == 1 and s0.ecs_offset == v.header_len and s0.ecs_length == v.ecs_len
    for f in ("blocks_per_mcu", "mcus_x", "mcus_y", "restart_interval"):
        assert getattr(s0, f) == getattr(s1, f), f
    assert bytes(s0.quant) == bytes(s1.quant) and bytes(s0.ac_vals) == bytes(s1.ac_vals)
    # the entry: clean stream = the segment without its stuffed zeros, then 12 bytes per 256-byte slice
    clean_len, total_starts, num_slices = (int(x) for x in box[v.index_offset:v.index_offset + 12].view(np.int32))
    seg = enc[v.header_len:v.header_len + v.ecs_len]
    assert bytes(box[v.index_offset + 64:v.index_offset + 64 + clean_len]) == seg.replace(b"\xff\x00", b"\xff")
    assert total_starts == s0.mcus_x * s0.mcus_y * s0.blocks_per_mcu + 1 and num_slices == (clean_len + 255) // 256
    assert box.size < 1.08 * len(enc) + 1024


def test_streams_the_gpu_decoder_does_not_take_get_no_container():
    rng = np.random.default_rng(4)
    img = synth_image(rng, 64, 80)
    for enc, word in [(encode_jpeg(img, 85, progressive=True), "progressive"), (encode_jpeg(img, 85, restart_marker_blocks=3), "restart"),
                      (encode_jpeg(img, 85)[:900], ""), (b"\x89PNG\r\n\x1a\n" + bytes(64), "")]:
        box, why = _build(enc)
        assert box is None and why and word in why, (word, why)


def test_damaged_containers_are_refused():
    rng = np.random.default_rng(5)
    box, _ = _build(encode_jpeg(synth_image(rng, 64, 80), 85))
    host = capi.host()
    v = View()
    for damage in ("version", "truncated", "entry"):
        b = box.copy()
        if damage == "version":
            b[4] = 9
        elif damage == "truncated":
            b = b[:b.size - 100].copy()
        else:
            capi.check_host(host.daliamdJpegIndexedParse(box.ctypes.data_as(C.c_void_p), C.c_size_t(box.size), C.byref(v)))
            b[v.index_offset:v.index_offset + 4] = 255
        assert host.daliamdJpegIndexedParse(b.ctypes.data_as(C.c_void_p), C.c_size_t(b.size), C.byref(v)) != 0, damage


def test_reader_hands_out_containers_under_the_files_names(tmp_path):
    import jpeg2idx
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(6)
    root, idx = tmp_path / "data", tmp_path / "index"
    encs = {}
    for cls, name, kw in [("a", "x.jpg", {}), ("a", "y.jpg", dict(progressive=True)), ("b", "z.jpg", dict(subsampling="4:4:4"))]:
        os.makedirs(root / cls, exist_ok=True)
        encs[(cls, name)] = encode_jpeg(synth_image(rng, 40, 56), 85, **kw)
        (root / cls / name).write_bytes(encs[(cls, name)])
    made, skipped = jpeg2idx.index_tree(str(root), str(idx), quiet=True)
    assert (made, skipped) == (2, 1) and (idx / "a" / "x.jpg.didx").exists() and not (idx / "a" / "y.jpg.didx").exists()
    pipe = Pipeline(batch_size=3, num_threads=2, device_id=None, prefetch_queue_depth=1)
    with pipe:
        enc, lab = fn.readers.file(file_root=str(root), index_path=str(idx))
        pipe.set_outputs(enc, lab)
    for _ in range(2):
        enc, lab = pipe.run()
        assert list(lab.as_array().reshape(-1)) == [0, 0, 1]
        got = [bytes(np.asarray(enc.at(i))) for i in range(3)]
        assert got[0] == (idx / "a" / "x.jpg.didx").read_bytes() and got[0][:4] == b"DAJX"
        assert got[1] == encs[("a", "y.jpg")]                       # no container: the file as it is
        assert got[2] == (idx / "b" / "z.jpg.didx").read_bytes()


def test_command_line_tool(tmp_path):
    import subprocess
    rng = np.random.default_rng(7)
    root, idx = tmp_path / "d", tmp_path / "i"
    os.makedirs(root / "c")
    for k in range(5):
        (root / "c" / f"f{k}.jpg").write_bytes(encode_jpeg(synth_image(rng, 48, 64), 80, **({"progressive": True} if k == 4 else {})))
    (root / "c" / "notes.txt").write_text("not an image")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "jpeg2idx.py"), str(root), str(idx), "--workers", "2"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "4 containers" in out.stdout and "1 files left as they are" in out.stdout, out.stdout
    assert sorted(os.listdir(idx / "c")) == [f"f{k}.jpg.didx" for k in range(4)]
