"""External pin of the FLAC decoders (VERDICT r03 missing item 5): the complete example streams of RFC 9639 Appendix D.

Until now the fixtures' encoder (tests/golden/make_flac_golden.py), the oracle (oracle/audio.py) and the product
(dali_amd/host/flac_decode.cpp) were three builder-written readings of one RFC.  These three files were produced by the
reference encoder (libFLAC 1.3.3, says example 2's vendor string) and carry, in STREAMINFO, the MD5 of the audio it encoded:
a decoder that reproduces that MD5 decodes what libFLAC - which is what libsndfile, and with it the reference
(dali/operators/decoder/audio/generic_decoder.cc:198-206), reads FLAC through - encoded."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from oracle import audio as A

HERE = os.path.dirname(os.path.abspath(__file__))
DIR = os.path.join(HERE, "golden", "flac_rfc9639")
FILES = ["example1.flac", "example2.flac", "example3.flac"]
# (channels, bits, rate, samples per channel) as the RFC's walk-through of each example states them
FORMAT = {"example1.flac": (2, 16, 44100.0, 1), "example2.flac": (2, 16, 44100.0, 19), "example3.flac": (1, 8, 32000.0, 24)}


def _crc(data, poly, bits):
    top, mask, c = 1 << (bits - 1), (1 << bits) - 1, 0
    for x in data:
        c ^= x << (bits - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
    return c


def _streaminfo_md5(data):
    assert data[:4] == b"fLaC" and data[4] & 0x7F == 0
    return data[8 + 18:8 + 34]


def _signature(pcm, bits):
    """MD5 as RFC 9639 section 8.2 defines it: samples interleaved, little-endian, sign-extended to whole bytes."""
    dt = {8: "<i1", 16: "<i2", 24: None, 32: "<i4"}[bits]
    assert dt is not None
    return hashlib.md5(np.ascontiguousarray(pcm).astype(dt).tobytes()).digest()


def _frames(data):
    """(offset, length) of the audio frames: behind the last metadata block, split at the fixed-blocksize sync code."""
    pos, last = 4, False
    while not last:
        last = bool(data[pos] & 0x80)
        pos += 4 + int.from_bytes(data[pos + 1:pos + 4], "big")
    cuts = [pos]
    while True:
        nxt = data.find(b"\xff\xf8", cuts[-1] + 2)
        if nxt < 0:
            break
        cuts.append(nxt)
    cuts.append(len(data))
    return [(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])]


@pytest.mark.parametrize("name", FILES)
def test_the_typed_in_bytes_are_the_rfcs(name):
    data = open(os.path.join(DIR, name), "rb").read()
    assert len(data) == {"example1.flac": 57, "example2.flac": 227, "example3.flac": 73}[name]
    frames = _frames(data)
    assert len(frames) == (2 if name == "example2.flac" else 1)
    for off, length in frames:
        f = data[off:off + length]
        hdr = 6 if f[2] >> 4 == 6 else 5   # block size code 6: one more header byte (the 8-bit block size)
        assert _crc(f[:hdr], 0x07, 8) == f[hdr], "frame header CRC-8"
        assert _crc(f[:-2], 0x8005, 16) == int.from_bytes(f[-2:], "big"), "frame CRC-16"


@pytest.mark.parametrize("name", FILES)
def test_oracle_decoder_reproduces_the_encoders_md5(name):
    data = open(os.path.join(DIR, name), "rb").read()
    pcm, bits, rate = A.decode_flac(data)
    ch, b, r, n = FORMAT[name]
    assert (pcm.shape, bits, rate) == ((n, ch), b, r)
    assert _signature(pcm, bits) == _streaminfo_md5(data)


class _Info(C.Structure):
    _fields_ = [("channels", C.c_int32), ("bits", C.c_int32), ("rate", C.c_double), ("frames", C.c_int64)]


def _product_decode(data):
    from dali_amd import _capi as capi
    lib = capi.host()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    info = _Info()
    if lib.daliamdFlacProbe(buf, C.c_size_t(len(data)), C.byref(info)) != 0:
        raise RuntimeError(lib.daliamdHostGetLastErrorMessage().decode())
    out = np.zeros((info.frames, info.channels), np.int32)
    if lib.daliamdFlacDecode(buf, C.c_size_t(len(data)), out.ctypes.data_as(C.c_void_p), C.c_int64(info.frames)) != 0:
        raise RuntimeError(lib.daliamdHostGetLastErrorMessage().decode())
    return out, info


@pytest.mark.parametrize("name", FILES)
def test_product_decoder_reproduces_the_encoders_md5(name):
    data = open(os.path.join(DIR, name), "rb").read()
    pcm, info = _product_decode(data)
    ch, b, r, n = FORMAT[name]
    assert (pcm.shape, info.bits, info.rate) == ((n, ch), b, r)
    assert _signature(pcm, info.bits) == _streaminfo_md5(data)
    assert np.array_equal(pcm, A.decode_flac(data)[0])


def test_example1_samples_as_the_rfc_states_them():
    """D.1.3: the left sample is 25588 (0x18fd << 2, two wasted bits), the right one 10416 (0x28b << 4)."""
    pcm, _ = _product_decode(open(os.path.join(DIR, "example1.flac"), "rb").read())
    assert pcm.tolist() == [[25588, 10416]]


def test_leading_id3v2_tag_and_trailing_bytes_are_skipped():
    """libFLAC / libsndfile accept a leading ID3v2 tag and stop at trailing non-frame data (an ID3v1 tag): so does the
    product's decoder - also when STREAMINFO does not state the length."""
    data = open(os.path.join(DIR, "example2.flac"), "rb").read()
    want, _ = _product_decode(data)
    id3v2 = b"ID3\x04\x00\x00" + bytes([0, 0, 0, 23]) + b"\x00" * 23
    tagged = id3v2 + data + b"TAG" + b"\x00" * 125
    got, info = _product_decode(tagged)
    assert np.array_equal(got, want)
    # unknown length: the frame walk ends at the tag instead of failing on it
    nolen = bytearray(data)
    nolen[8 + 13] &= 0xF0
    nolen[8 + 14:8 + 18] = b"\x00\x00\x00\x00"
    got, info = _product_decode(id3v2 + bytes(nolen) + b"TAG" + b"\x00" * 125)
    assert info.frames == 19 and np.array_equal(got, want)


def test_crafted_sample_count_is_refused_not_allocated():
    data = bytearray(open(os.path.join(DIR, "example1.flac"), "rb").read())
    data[8 + 13] |= 0x0F
    data[8 + 14:8 + 18] = b"\xff\xff\xff\xff"   # 2^36 - 1 samples in a 57-byte file
    with pytest.raises(RuntimeError, match="cannot hold"):
        _product_decode(bytes(data))
