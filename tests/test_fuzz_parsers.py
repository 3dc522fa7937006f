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
