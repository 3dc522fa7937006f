"""decoders.audio beyond 16-bit WAV (the reference reads everything libsndfile reads, generic_decoder.cc:170-206; LibriSpeech
ships as FLAC): FLAC streams and 8- / 24- / 32-bit PCM WAV.

FLAC is lossless integer work: the committed fixtures (tests/golden/flac/*.flac + the PCM they were made from, written by
tests/golden/make_flac_golden.py) must decode bit for bit - by the oracle's plain-Python decoder and by the product's C++ one
(ctypes and through fn.decoders.audio with every output type, libsndfile's conversions).  No FLAC tool exists in this image:
the fixtures' encoder, the oracle and the product are three separate pieces of code written from RFC 9639."""
import ctypes as C
import glob
import io
import os
import struct

import numpy as np
import pytest

from oracle import audio as A

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "flac", "*.flac")))


def _truth(path):
    return np.load(path[:-5].replace("nolength", "mono16_fixed") + ".npz")["pcm"]


def test_fixture_set_is_complete():
    assert [os.path.basename(f)[:-5] for f in FIXTURES] == ["lpc24", "mono16_fixed", "nolength", "pcm8_const_verb",
                                                           "stereo16_modes", "wasted_escape"]


@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[:-5])
def test_oracle_flac_decoder_on_the_fixtures(path):
    pcm, bits, rate = A.decode_flac(open(path, "rb").read())
    assert np.array_equal(pcm, _truth(path))
    assert (bits, rate) == {"lpc24": (24, 48000.0), "pcm8_const_verb": (8, 8000.0), "stereo16_modes": (16, 44100.0)}.get(
        os.path.basename(path)[:-5], (16, 16000.0))


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


@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[:-5])
def test_product_flac_decoder_on_the_fixtures(path):
    data = open(path, "rb").read()
    pcm, info = _product_decode(data)
    ref, bits, rate = A.decode_flac(data)
    assert np.array_equal(pcm, _truth(path)) and np.array_equal(pcm, ref)
    assert (info.bits, info.rate, info.channels) == (bits, rate, ref.shape[1])


def test_corrupt_flac_frames_are_refused():
    data = bytearray(open(FIXTURES[1], "rb").read())          # mono16_fixed
    broken = bytearray(data)
    broken[len(broken) // 2] ^= 0x10                           # a residual bit: the frame's CRC-16 no longer matches
    with pytest.raises(RuntimeError, match="FLAC: broken frame"):
        _product_decode(bytes(broken))
    with pytest.raises(RuntimeError, match="FLAC"):
        _product_decode(bytes(data[:len(data) // 2]))          # truncated: fewer samples than STREAMINFO promises
    with pytest.raises(RuntimeError, match="not a FLAC stream"):
        _product_decode(b"fLaX" + bytes(data[4:]))


def _wav(samples, bits, rate=16000, channels=1):
    x = np.asarray(samples)
    if bits == 8:
        raw = (x + 128).astype(np.uint8).tobytes()
    elif bits == 16:
        raw = x.astype("<i2").tobytes()
    elif bits == 24:
        v = x.astype(np.int64) & 0xFFFFFF
        raw = np.stack([v & 255, (v >> 8) & 255, v >> 16], -1).astype(np.uint8).tobytes()
    else:
        raw = x.astype("<i4").tobytes()
    b = io.BytesIO()
    b.write(b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE")
    b.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, channels, rate, rate * channels * bits // 8, channels * bits // 8, bits))
    b.write(b"data" + struct.pack("<I", len(raw)) + raw)
    return b.getvalue()


def _decode_pipe(encoded, **kw):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    pipe = Pipeline(batch_size=len(encoded), num_threads=2, device_id=None, prefetch_queue_depth=1)
    with pipe:
        enc = fn.external_source(name="enc")
        pipe.set_outputs(*fn.decoders.audio(enc, **kw))
    pipe.build()
    pipe.feed_input("enc", [np.frombuffer(e, np.uint8) for e in encoded])
    return pipe.run()


@pytest.mark.parametrize("dtype", ["FLOAT", "INT16", "INT32"])
def test_decoders_audio_reads_flac_and_every_pcm_width(dtype):
    """Output types as libsndfile's reads give them: float = x / 2^(bits-1), int16 = the top 16 bits, int32 = x << (32 - bits)."""
    from dali_amd import types
    rng = np.random.default_rng(8)
    items = []        # (encoded, integer samples [frames][channels], bits, rate)
    for f in FIXTURES:
        data = open(f, "rb").read()
        pcm, bits, rate = A.decode_flac(data)
        items.append((data, pcm, bits, rate))
    for bits in (8, 16, 24, 32):
        x = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), (300, 2))
        items.append((_wav(x, bits, 22050, 2), x.astype(np.int64).astype(np.int32), bits, 22050.0))
    audio, rate = _decode_pipe([it[0] for it in items], dtype=getattr(types, dtype))
    conv = {"FLOAT": A.pcm_to_float, "INT16": A.pcm_to_int16, "INT32": A.pcm_to_int32}[dtype]
    for i, (_, pcm, bits, sr) in enumerate(items):
        want = conv(pcm, bits)
        want = want[:, 0] if want.shape[1] == 1 else want
        got = audio.at(i)
        assert got.dtype == want.dtype and got.shape == want.shape, (i, got.shape, want.shape)
        assert np.array_equal(got, want), i
        assert float(rate.at(i)) == sr


def test_flac_downmix_and_resampling_go_through_the_float_path():
    data = open([f for f in FIXTURES if "stereo16" in f][0], "rb").read()
    pcm, bits, rate = A.decode_flac(data)
    audio, _ = _decode_pipe([data], downmix=True)
    f = A.pcm_to_float(pcm, bits)
    want = f[:, 0] * np.float32(0.5) + f[:, 1] * np.float32(0.5)          # downmixing.h:50-76, equal weights in channel order
    assert np.array_equal(audio.at(0), want)
    audio, sr = _decode_pipe([data], downmix=True, sample_rate=16000.0)
    assert float(sr.at(0)) == 16000.0 and audio.at(0).shape == (int(np.ceil(len(pcm) * 16000.0 / rate)),)


def test_ogg_is_refused_with_a_message():
    with pytest.raises(RuntimeError, match="Ogg streams are not supported"):
        _decode_pipe([b"OggS" + bytes(100)])
