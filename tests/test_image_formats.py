"""Host decoders of the loss-less container formats (PNG, BMP, PNM) against Pillow: every 8-bit source must give exactly
Image.convert("RGB"); windows must equal the crop of the full decode; broken streams must be refused with a message."""
import ctypes as C
import io
import struct
import zlib

import numpy as np
import pytest
from PIL import Image

from dali_amd import _capi as capi
from tests.util import synth_image


def probe(data):
    lib = capi.host()
    buf = np.frombuffer(data, np.uint8)
    fmt, w, h = C.c_int(), C.c_int32(), C.c_int32()
    rc = lib.daliamdImageProbe(buf.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)), C.byref(fmt), C.byref(w), C.byref(h))
    if rc:
        raise RuntimeError(lib.daliamdHostGetLastErrorMessage().decode())
    return fmt.value, h.value, w.value


def decode(data, window=None, pad=0):
    lib = capi.host()
    buf = np.frombuffer(data, np.uint8)
    _, H, W = probe(data)
    y0, x0, h, w = window or (0, 0, H, W)
    pitch = 3 * w + pad
    out = np.full((h, pitch), 0xA5, np.uint8)
    rc = lib.daliamdImageDecodeRgb(buf.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)), out.ctypes.data_as(C.c_void_p),
                                   C.c_int64(pitch), y0, x0, h if window else 0, w if window else 0)
    if rc:
        raise RuntimeError(lib.daliamdHostGetLastErrorMessage().decode())
    assert (out[:, 3 * w:] == 0xA5).all(), "bytes behind the row were touched"
    return out[:, :3 * w].reshape(h, w, 3)


def encode(img, fmt, **kw):
    b = io.BytesIO()
    img.save(b, fmt, **kw)
    return b.getvalue()


def pil_rgb(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


RNG = np.random.default_rng(3)
RGB = synth_image(RNG, 37, 53)
GRAY = synth_image(RNG, 29, 41, 1).reshape(29, 41)


def png_cases():
    rgb, gray = Image.fromarray(RGB), Image.fromarray(GRAY)
    rgba = Image.fromarray(np.dstack([RGB, RNG.integers(0, 256, RGB.shape[:2], dtype=np.uint8)]))
    la = Image.fromarray(np.dstack([GRAY, 255 - GRAY]), "LA")
    pal = rgb.quantize(200)
    pal16 = rgb.quantize(16)
    pal2 = rgb.quantize(2)
    one = gray.point(lambda v: 255 * (v > 128)).convert("1")
    yield "rgb", encode(rgb, "PNG")
    yield "rgb-level0", encode(rgb, "PNG", compress_level=0)
    yield "rgb-interlaced", _interlace(RGB)
    yield "gray", encode(gray, "PNG")
    yield "gray-interlaced", _interlace(GRAY[..., None])
    yield "rgba", encode(rgba, "PNG")
    yield "gray-alpha", encode(la, "PNG")
    yield "palette-8", encode(pal, "PNG")
    yield "palette-4", encode(pal16, "PNG", bits=4)
    yield "palette-1", encode(pal2, "PNG", bits=1)
    yield "bilevel", encode(one, "PNG")
    yield "one-pixel", encode(Image.fromarray(RGB[:1, :1]), "PNG")
    yield "narrow", encode(Image.fromarray(RGB[:, :1]), "PNG")


def _chunk(kind, body):
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))


def _interlace(arr):
    """Adam7 PNG written by hand (Pillow cannot write interlaced files): every pass with a different filter type."""
    h, w, c = arr.shape
    passes = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
    raw = bytearray()
    for k, (x0, y0, dx, dy) in enumerate(passes):
        sub = arr[y0::dy, x0::dx]
        if sub.size == 0:
            continue
        prev = np.zeros(sub.shape[1] * c, np.int32)
        for row in sub.reshape(sub.shape[0], -1).astype(np.int32):
            ft = k % 5
            left = np.concatenate([np.zeros(c, np.int32), row[:-c]])
            upleft = np.concatenate([np.zeros(c, np.int32), prev[:-c]])
            if ft == 0:
                f = row
            elif ft == 1:
                f = row - left
            elif ft == 2:
                f = row - prev
            elif ft == 3:
                f = row - ((left + prev) >> 1)
            else:
                p = left + prev - upleft
                pa, pb, pc = abs(p - left), abs(p - prev), abs(p - upleft)
                pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
                f = row - pred
            raw += bytes([ft]) + (f & 255).astype(np.uint8).tobytes()
            prev = row
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 2 if c == 3 else 0, 0, 0, 1)
    return b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", zlib.compress(bytes(raw))) + _chunk(b"IEND", b"")


@pytest.mark.parametrize("name,data", list(png_cases()))
def test_png_matches_pillow(name, data):
    fmt, h, w = probe(data)
    ref = pil_rgb(data)
    assert fmt == 2 and (h, w) == ref.shape[:2]
    assert np.array_equal(decode(data, pad=5), ref)
    if h > 8 and w > 8:
        assert np.array_equal(decode(data, window=(3, 5, h - 7, w - 6)), ref[3:h - 4, 5:w - 1])


def test_png_sixteen_bit_keeps_the_high_byte():
    v = (np.arange(20 * 30, dtype=np.uint32).reshape(20, 30) * 109 % 65536).astype(">u2")
    raw = b"".join(b"\0" + row.tobytes() for row in v)
    data = (b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 30, 20, 16, 0, 0, 0, 0)) +
            _chunk(b"IDAT", zlib.compress(raw)) + _chunk(b"IEND", b""))
    got = decode(data)
    assert np.array_equal(got[..., 0], (v.astype(np.uint16) >> 8).astype(np.uint8)) and (got[..., 0] == got[..., 2]).all()


def bmp_cases():
    rgb = Image.fromarray(RGB)
    yield "24-bit", encode(rgb, "BMP")
    yield "8-bit-palette", encode(rgb.quantize(256), "BMP")
    yield "8-bit-gray", encode(Image.fromarray(GRAY), "BMP")
    yield "1-bit", encode(Image.fromarray(GRAY).point(lambda v: 255 * (v > 128)).convert("1"), "BMP")
    yield "32-bit", encode(Image.fromarray(np.dstack([RGB, np.full(RGB.shape[:2], 255, np.uint8)])), "BMP")
    # top-down variant of the 24-bit file: negative height, rows in natural order
    d = bytearray(encode(rgb, "BMP"))
    off, h, stride = struct.unpack_from("<I", d, 10)[0], RGB.shape[0], (RGB.shape[1] * 3 + 3) // 4 * 4
    rows = [bytes(d[off + r * stride: off + (r + 1) * stride]) for r in range(h)]
    struct.pack_into("<i", d, 22, -h)
    d[off:] = b"".join(reversed(rows))
    yield "24-bit-top-down", bytes(d)


@pytest.mark.parametrize("name,data", list(bmp_cases()))
def test_bmp_matches_pillow(name, data):
    fmt, h, w = probe(data)
    ref = pil_rgb(data)
    assert fmt == 3 and (h, w) == ref.shape[:2]
    assert np.array_equal(decode(data, pad=1), ref)
    assert np.array_equal(decode(data, window=(2, 3, h - 5, w - 9)), ref[2:h - 3, 3:w - 6])


def pnm_cases():
    yield "P6", encode(Image.fromarray(RGB), "PPM")
    yield "P5", encode(Image.fromarray(GRAY), "PPM")
    yield "P4", encode(Image.fromarray(GRAY).point(lambda v: 255 * (v > 128)).convert("1"), "PPM")
    h, w = 5, 7
    small = RGB[:h, :w]
    yield "P3-with-comments", (f"P3\n# a comment\n{w} {h}\n# another\n255\n" +
                               "\n".join(" ".join(str(v) for v in row.reshape(-1)) for row in small) + "\n").encode()
    yield "P2-maxval-7", (f"P2 {w} {h} 7\n" + " ".join(str(int(v) % 8) for v in GRAY[:h, :w].reshape(-1))).encode()
    yield "P1", (f"P1\n{w} {h}\n" + "".join(str(int(v) & 1) for v in GRAY[:h, :w].reshape(-1))).encode()


@pytest.mark.parametrize("name,data", list(pnm_cases()))
def test_pnm_matches_pillow(name, data):
    fmt, h, w = probe(data)
    ref = pil_rgb(data)
    assert fmt == 4 and (h, w) == ref.shape[:2]
    assert np.array_equal(decode(data), ref)
    assert np.array_equal(decode(data, window=(1, 2, h - 2, w - 3)), ref[1:h - 1, 2:w - 1])


def test_broken_streams_are_refused():
    png = encode(Image.fromarray(RGB), "PNG")
    flipped = bytearray(png)
    flipped[60] ^= 0x40
    with pytest.raises(RuntimeError, match="CRC error"):
        decode(bytes(flipped))
    with pytest.raises(RuntimeError, match="chunk exceeds the stream|truncated"):
        decode(png[:len(png) // 2])
    with pytest.raises(RuntimeError, match="unrecognised image format"):
        probe(b"GIF89a" + bytes(40))
    bmp = encode(Image.fromarray(RGB), "BMP")
    with pytest.raises(RuntimeError, match="truncated pixel data"):
        decode(bmp[:200])
    with pytest.raises(RuntimeError, match="does not fit"):
        decode(bmp, window=(30, 0, 10, 10))
    rle = bytearray(bmp)
    struct.pack_into("<I", rle, 30, 1)
    with pytest.raises(RuntimeError, match="not supported"):
        decode(bytes(rle))
    assert probe(b"\xff\xd8\xff\xe0" + bytes(20))[0] == 1      # JPEG: only identified here
