"""daliamdJpegEncodeBaselineScan: decoded coefficients written out again as ONE sequential baseline scan with the Annex K
tables (the lossless re-encoding `jpegtran` performs).  decoders.image(mixed) keeps progressive / multi-scan streams
resident in the encoded-stream cache in that form, so the device decodes them from the second epoch on.

Pin: libjpeg-turbo.  The re-encoded segment, wrapped into a complete file (the original's DQT segments and frame header,
the DHT the scan analysis reports, one SOS), must decode with Pillow to the pixels of the original file - which also proves
the embedded Annex K tables, code for code, because libjpeg-turbo builds its decoder from the DHT the file carries and a
default-encoded file's DHT is compared with them directly."""
import ctypes as C
import io
import struct

import numpy as np
import pytest
from PIL import Image

from dali_amd import _capi as capi
from tests.util import encode_jpeg, synth_image


def _segments(data):
    """[(marker, payload bytes)] of the headers up to SOS."""
    out, p = [], 2
    while p < len(data):
        assert data[p] == 0xFF
        m = data[p + 1]
        n = struct.unpack(">H", data[p + 2:p + 4])[0]
        out.append((m, data[p + 4:p + 2 + n]))
        if m == 0xDA:
            break
        p += 2 + n
    return out


def _transcode(enc):
    host = capi.host()
    buf = np.frombuffer(enc, np.uint8)
    info = capi.JpegInfo()
    capi.check_host(host.daliamdJpegParse(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), C.byref(info)))
    nc = info.num_components
    coefs = [np.zeros(int(info.coef_elems[c]), np.int16) for c in range(nc)]
    ptrs = (C.c_void_p * 4)(*[c.ctypes.data for c in coefs], *([None] * (4 - nc)))
    quant = np.zeros((4, 64), np.uint16)
    capi.check_host(host.daliamdJpegDecodeCoefficients(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), C.byref(info), ptrs,
                                                       quant.ctypes.data_as(C.c_void_p)))
    out = np.zeros(2 * len(enc) + 4096, np.uint8)
    length = C.c_size_t(0)
    scan = capi.JpegScan()
    capi.check_host(host.daliamdJpegEncodeBaselineScan(C.byref(info), ptrs, quant.ctypes.data_as(C.c_void_p),
                                                       out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size), C.byref(length),
                                                       C.byref(scan)))
    return info, scan, bytes(out[:length.value])


def _wrap(enc, info, scan, ecs):
    """A complete baseline file around the re-encoded segment."""
    segs = _segments(enc)
    f = bytearray(b"\xff\xd8")
    for m, payload in segs:
        if m == 0xDB or 0xE0 <= m <= 0xEF:                 # quantisation tables, APPn (JFIF / Adobe colour transform)
            f += bytes([0xFF, m]) + struct.pack(">H", len(payload) + 2) + payload
    sof = next(p for m, p in segs if m in (0xC0, 0xC1, 0xC2))
    f += b"\xff\xc0" + struct.pack(">H", len(sof) + 2) + sof
    nc = info.num_components
    for cls, bits_all, vals_all in ((0, scan.dc_bits, scan.dc_vals), (1, scan.ac_bits, scan.ac_vals)):
        for t in range(2 if nc == 3 else 1):
            bits = bytes(bits_all[t])
            vals = bytes(vals_all[t])[:sum(bits)]
            f += b"\xff\xc4" + struct.pack(">H", 3 + 16 + len(vals)) + bytes([(cls << 4) | t]) + bits + vals
    comp_ids = [sof[6 + 3 * c] for c in range(nc)]
    sos = bytes([nc]) + b"".join(bytes([comp_ids[c], (scan.dc_sel[c] << 4) | scan.ac_sel[c]]) for c in range(nc)) + b"\x00\x3f\x00"
    f += b"\xff\xda" + struct.pack(">H", len(sos) + 2) + sos + ecs + b"\xff\xd9"
    return bytes(f)


CASES = [((97, 131), "4:2:0", dict(progressive=True)), ((240, 320), "4:4:4", dict(progressive=True)),
         ((75, 211), "4:2:2", dict(progressive=True)), ((200, 150), "4:2:0", dict(progressive=True, optimize=True)),
         ((64, 48), "4:2:0", {}), ((333, 500), "4:2:0", dict(optimize=True)), ((50, 70), "gray", dict(progressive=True)),
         ((17, 9), "4:2:0", dict(progressive=True)), ((8, 8), "4:4:4", dict(progressive=True))]


@pytest.mark.parametrize("hw,sub,kw", CASES)
def test_reencoded_stream_decodes_to_the_pixels_of_the_original(hw, sub, kw):
    rng = np.random.default_rng(sum(hw))
    img = synth_image(rng, *hw, 1 if sub == "gray" else 3)
    enc = encode_jpeg(img, 90 if hw[0] % 2 else 60, **({} if sub == "gray" else {"subsampling": sub}), **kw)
    info, scan, ecs = _transcode(enc)
    assert scan.eligible == 1 and scan.ecs_length == len(ecs) and scan.restart_interval == 0
    again = _wrap(enc, info, scan, ecs)
    im = Image.open(io.BytesIO(again))
    assert im.info.get("progressive", 0) == 0
    assert np.array_equal(np.asarray(im), np.asarray(Image.open(io.BytesIO(enc))))
    # ... and the product's own analysis of the wrapped file finds an eligible baseline stream with the same structure
    host = capi.host()
    buf = np.frombuffer(again, np.uint8)
    info2, scan2 = capi.JpegInfo(), capi.JpegScan()
    capi.check_host(host.daliamdJpegParse(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), C.byref(info2)))
    capi.check_host(host.daliamdJpegAnalyzeScan(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), C.byref(info2), C.byref(scan2)))
    assert scan2.eligible == 1 and scan2.ecs_length == len(ecs)
    for f in ("blocks_per_mcu", "mcus_x", "mcus_y"):
        assert getattr(scan, f) == getattr(scan2, f), f
    for f in ("comp_of_block", "h_of_block", "v_of_block", "dc_sel", "ac_sel"):
        assert bytes(getattr(scan, f)) == bytes(getattr(scan2, f)), f
    assert bytes(scan.quant) == bytes(scan2.quant) and bytes(scan.dc_bits) == bytes(scan2.dc_bits)
    assert bytes(scan.ac_vals) == bytes(scan2.ac_vals) and bytes(scan.dc_vals) == bytes(scan2.dc_vals)


def test_embedded_tables_are_the_ones_libjpeg_writes_by_default():
    rng = np.random.default_rng(4)
    plain = encode_jpeg(synth_image(rng, 40, 56), 75)                  # default tables: T.81 Annex K.3
    prog = encode_jpeg(synth_image(rng, 40, 56), 75, progressive=True)
    host = capi.host()
    buf = np.frombuffer(plain, np.uint8)
    info, want = capi.JpegInfo(), capi.JpegScan()
    capi.check_host(host.daliamdJpegAnalyzeHeader(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), C.byref(info), C.byref(want)))
    _, got, _ = _transcode(prog)
    for f in ("dc_bits", "dc_vals", "ac_bits", "ac_vals", "dc_sel", "ac_sel"):
        assert bytes(getattr(got, f)) == bytes(getattr(want, f)), f


def test_a_too_small_buffer_is_an_error_not_an_overrun():
    rng = np.random.default_rng(5)
    enc = encode_jpeg(synth_image(rng, 120, 160), 90, progressive=True)
    host = capi.host()
    buf = np.frombuffer(enc, np.uint8)
    info = capi.JpegInfo()
    capi.check_host(host.daliamdJpegParse(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), C.byref(info)))
    coefs = [np.zeros(int(info.coef_elems[c]), np.int16) for c in range(3)]
    ptrs = (C.c_void_p * 4)(*[c.ctypes.data for c in coefs], None)
    quant = np.zeros((4, 64), np.uint16)
    capi.check_host(host.daliamdJpegDecodeCoefficients(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), C.byref(info), ptrs,
                                                       quant.ctypes.data_as(C.c_void_p)))
    out = np.full(1000 + 64, 0xAB, np.uint8)
    length, scan = C.c_size_t(0), capi.JpegScan()
    rc = host.daliamdJpegEncodeBaselineScan(C.byref(info), ptrs, quant.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                            C.c_size_t(1000), C.byref(length), C.byref(scan))
    assert rc != 0 and b"too small" in host.daliamdHostGetLastErrorMessage()
    assert (out[1000:] == 0xAB).all()
