"""Smoke fuzzing of every parser that sees untrusted bytes (JPEG header / scan analysis / host entropy decoder, PNG,
BMP, PNM): mutated streams must be decoded or refused with an error - never crash, never report a segment outside
the stream.  (The same drivers run for minutes under AddressSanitizer + UBSan during development: that is how the
unmasked DC category in the host entropy decoder was found.)"""
import ctypes as C
import io

import numpy as np
import pytest
from PIL import Image

from dali_amd import _capi as capi
from tests.util import encode_jpeg, synth_image


def _mutations(seed, rng, count):
    for _ in range(count):
        d = bytearray(seed)
        if rng.integers(0, 4) == 0:
            d = d[:int(rng.integers(0, len(d) + 1))]
        for _ in range(int(rng.integers(1, 5))):
            if not d:
                break
            pos = int(rng.integers(0, min(len(d), 600))) if rng.integers(0, 2) else int(rng.integers(0, len(d)))
            d[pos] = int(rng.integers(0, 256)) if rng.integers(0, 2) else d[pos] ^ (1 << int(rng.integers(0, 8)))
        yield bytes(d)


def test_jpeg_host_parsers_survive_mutations():
    host = capi.host()
    rng = np.random.default_rng(99)
    seeds = [encode_jpeg(synth_image(rng, 37, 53), 85, subsampling="4:2:0"),
             encode_jpeg(synth_image(rng, 40, 40), 85, subsampling="4:2:0", progressive=True),
             encode_jpeg(synth_image(rng, 33, 47), 85, subsampling="4:2:2", restart_marker_blocks=3),
             encode_jpeg(synth_image(rng, 30, 30, 1), 80)]
    decoded = refused = 0
    for seed in seeds:
        for data in _mutations(seed, rng, 400):
            buf = np.frombuffer(data, np.uint8) if data else np.zeros(1, np.uint8)
            info = capi.JpegInfo()
            if host.daliamdJpegParse(buf.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)), C.byref(info)) != 0 or \
                    info.num_components not in (1, 3):
                refused += 1
                continue
            elems = [int(info.coef_elems[c]) for c in range(info.num_components)]
            assert all(0 <= e <= 1 << 24 for e in elems)
            scan = capi.JpegScan()
            if host.daliamdJpegAnalyzeScan(buf.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)), C.byref(info),
                                           C.byref(scan)) == 0 and scan.eligible:
                assert 0 <= scan.ecs_offset and 0 <= scan.ecs_length and scan.ecs_offset + scan.ecs_length <= len(data)
            coef = np.zeros(sum(elems) + 1, np.int16)
            ptrs = (C.c_void_p * 4)()
            off = 0
            for c, e in enumerate(elems):
                ptrs[c] = coef.ctypes.data + 2 * off
                off += e
            quant = np.zeros(4 * 64, np.uint16)
            rc = host.daliamdJpegDecodeCoefficients(buf.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)), C.byref(info), ptrs,
                                                    quant.ctypes.data_as(C.c_void_p))
            decoded += rc == 0
            refused += rc != 0
    assert decoded > 100 and refused > 100


def test_raster_decoders_survive_mutations():
    host = capi.host()
    rng = np.random.default_rng(5)
    a = synth_image(rng, 31, 45)

    def enc(img, fmt, **kw):
        b = io.BytesIO()
        img.save(b, fmt, **kw)
        return b.getvalue()
    seeds = [enc(Image.fromarray(a), "PNG"), enc(Image.fromarray(a).quantize(16), "PNG", bits=4),
             enc(Image.fromarray(a), "BMP"), enc(Image.fromarray(a).quantize(256), "BMP"), enc(Image.fromarray(a), "PPM"),
             ("P3\n5 4\n255\n" + " ".join(str(int(v)) for v in a[:4, :5].reshape(-1))).encode()]
    decoded = refused = 0
    for seed in seeds:
        for data in _mutations(seed, rng, 400):
            buf = np.frombuffer(data, np.uint8) if data else np.zeros(1, np.uint8)
            fmt, w, h = C.c_int(), C.c_int32(), C.c_int32()
            if host.daliamdImageProbe(buf.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)), C.byref(fmt), C.byref(w),
                                      C.byref(h)) != 0 or fmt.value == 1 or w.value * h.value > 1 << 22:
                refused += 1
                continue
            out = np.zeros((h.value, 3 * w.value + 8), np.uint8)
            rc = host.daliamdImageDecodeRgb(buf.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)),
                                            out.ctypes.data_as(C.c_void_p), C.c_int64(out.shape[1]), 0, 0, 0, 0)
            assert (out[:, 3 * w.value:] == 0).all()
            decoded += rc == 0
            refused += rc != 0
    assert decoded > 100 and refused > 100


def test_exif_ifd_offset_cannot_wrap():
    """A 32-bit IFD offset of 0xFFFFFFFE used to pass `off + 2 > n` by wrap-around and read 4 GB past the APP1
    payload.  The parser must ignore such an Exif block (orientation stays 1) - and not crash."""
    host = capi.host()
    rng = np.random.default_rng(3)
    good = encode_jpeg(synth_image(rng, 16, 16), 85)
    for order, off in ((b"MM\x00*", b"\xff\xff\xff\xfe"), (b"II*\x00", b"\xfe\xff\xff\xff"),
                       (b"MM\x00*", b"\xff\xff\xff\xf2"), (b"MM\x00*", b"\x7f\xff\xff\xff")):
        payload = b"Exif\x00\x00" + order + off + b"\x00" * 8
        app1 = b"\xff\xe1" + (len(payload) + 2).to_bytes(2, "big") + payload
        for data in (b"\xff\xd8" + app1, good[:2] + app1 + good[2:]):
            buf = np.frombuffer(data, np.uint8)
            info = capi.JpegInfo()
            rc = host.daliamdJpegParse(buf.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)), C.byref(info))
            if len(data) > 100:
                assert rc == 0 and info.orientation == 1
    # an entry table that runs past the payload is ignored as well
    payload = b"Exif\x00\x00MM\x00*\x00\x00\x00\x08\xff\xff" + b"\x00" * 4
    data = good[:2] + b"\xff\xe1" + (len(payload) + 2).to_bytes(2, "big") + payload + good[2:]
    buf = np.frombuffer(data, np.uint8)
    info = capi.JpegInfo()
    assert host.daliamdJpegParse(buf.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)), C.byref(info)) == 0
    assert info.orientation == 1


def test_bmp_bitfield_masks_full_width_and_non_contiguous():
    """BI_BITFIELDS with a 32-bit wide channel mask used to divide by zero ((1u << 32) - 1 == 0 on x86); masks
    with holes are refused."""
    import struct
    host = capi.host()

    def bmp(masks, pixel):
        hdr = struct.pack("<IiiHHIIiiII", 40, 2, 1, 1, 32, 3, 8, 0, 0, 0, 0)
        body = hdr + struct.pack("<III", *masks) + struct.pack("<II", pixel, pixel)
        return b"BM" + struct.pack("<IHHI", 14 + len(body), 0, 0, 14 + 40 + 12) + body

    def decode(data):
        buf = np.frombuffer(data, np.uint8)
        fmt, w, h = C.c_int(), C.c_int32(), C.c_int32()
        if host.daliamdImageProbe(buf.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)), C.byref(fmt), C.byref(w),
                                  C.byref(h)) != 0:
            return None
        out = np.zeros((h.value, 3 * w.value), np.uint8)
        rc = host.daliamdImageDecodeRgb(buf.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)),
                                        out.ctypes.data_as(C.c_void_p), C.c_int64(out.shape[1]), 0, 0, 0, 0)
        return out if rc == 0 else None

    out = decode(bmp((0xFFFFFFFF, 0x0000FF00, 0x000000FF), 0x80FF4020))
    assert out is not None and out[0, 0] == 0x80FF4020 * 255 // 0xFFFFFFFF and out[0, 1] == 0x40 and out[0, 2] == 0x20
    out = decode(bmp((0xFF000000, 0x00FFFFFF, 0), 0x12FFFFFF))     # 24-bit wide mask: x * 255 needs 64 bits
    assert out is not None and out[0, 0] == 0x12 and out[0, 1] == 255
    assert decode(bmp((0x00FF00FF, 0x0000FF00, 0), 0)) is None      # mask with a hole
