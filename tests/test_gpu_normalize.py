"""GPU parity of fn.normalize (wave64 reductions in fp64 + element-wise pass) against the numpy oracle.
Tolerance: rtol = atol = 1e-3 like the reference's own test (dali/test/python/operator_1/test_normalize.py:209-211);
the fp64 accumulation keeps the real error around 1e-6."""
import numpy as np
import pytest

from oracle import normalize as ON
from tests.util import encode_jpeg, synth_image
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _run(samples, layout, **kw):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    pipe = Pipeline(batch_size=len(samples), num_threads=2, device_id=0, prefetch_queue_depth=1)
    with pipe:
        x = fn.external_source(source=lambda: [np.ascontiguousarray(b) for b in samples], batch=True, layout=layout)
        pipe.set_outputs(fn.normalize(x.gpu(), **kw))
    (out,) = pipe.run()
    assert "normalize" in pipe.executed_kernels()
    return [out[i].as_cpu() for i in range(len(samples))]


def _check(got, ref, rtol=1e-3, atol=1e-3):
    for i, (g, r) in enumerate(zip(got, ref)):
        assert g.shape == r.shape, (i, g.shape, r.shape)
        assert np.allclose(g, r, rtol=rtol, atol=atol), f"sample {i}: max err {np.abs(g - r).max()}"
        assert np.abs(g - r).max() < 2e-4 * max(1.0, np.abs(r).max()), "fp64 accumulation should do much better than 1e-3"


CASES = [
    (dict(), None),                                   # all axes
    (dict(axes=[0, 1]), (0, 1)),                      # per channel of an HWC image
    (dict(axis_names="HW"), (0, 1)),
    (dict(axes=[2]), (2,)),                           # per pixel over the channels
    (dict(axes=[1, 2]), (1, 2)),                      # per row
    (dict(axes=[0]), (0,)),                           # wide inner extent: per (x, c) column over the rows
    (dict(axes=[0, 1], ddof=1, epsilon=0.25), (0, 1)),
    (dict(axes=[0, 1], mean=100.0), (0, 1)),
    (dict(axes=[0, 1], stddev=50.0, epsilon=0.5), (0, 1)),
    (dict(mean=3.0, stddev=2.0, scale=4.0, shift=-1.0), None),
]


@pytest.mark.parametrize("kw,axes", CASES)
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_per_sample_normalization_matches_numpy(kw, axes, dtype):
    rng = np.random.default_rng(12)
    shapes = [(37, 53, 3), (120, 160, 3), (1, 7, 3), (300, 200, 3)]
    if dtype == np.uint8:
        batch = [synth_image(rng, h, w) for (h, w, _) in shapes]
    else:
        batch = [(rng.normal(5, 30, s) * np.linspace(0.5, 2, s[2])).astype(np.float32) for s in shapes]
    got = _run(batch, "HWC", **kw)
    okw = {"mean": kw.get("mean"), "stddev": kw.get("stddev"), "ddof": kw.get("ddof", 0), "eps": kw.get("epsilon", 0.0),
           "scale": kw.get("scale", 1.0), "shift": kw.get("shift", 0.0)}
    ref = [ON.normalize(b, axes, **okw).astype(np.float32) for b in batch]
    _check(got, ref)


def test_batch_normalization_per_channel():
    rng = np.random.default_rng(13)
    batch = [synth_image(rng, h, w) for (h, w) in [(40, 60), (100, 30), (64, 64)]]
    got = _run(batch, "HWC", axes=[0, 1], batch=True, ddof=1)
    _check(got, [r.astype(np.float32) for r in ON.normalize_batch(batch, (0, 1), ddof=1)])
    got = _run(batch, "HWC", batch=True)
    _check(got, [r.astype(np.float32) for r in ON.normalize_batch(batch, None)])


def test_spectrogram_style_per_frequency_normalisation_and_constant_rows():
    """[freq][time] float input normalised per frequency bin (axes=[1]); a constant row has zero variance -> zeros."""
    rng = np.random.default_rng(14)
    batch = [rng.normal(-40, 12, (513, t)).astype(np.float32) for t in (311, 1000, 64)]
    batch[0][7, :] = 3.5
    got = _run(batch, "ft", axes=[1])
    ref = [ON.normalize(b, (1,)).astype(np.float32) for b in batch]
    _check(got, ref)
    assert np.all(got[0][7] == 0)


def test_integer_output_with_scale_and_shift_saturates():
    from dali_amd import types
    rng = np.random.default_rng(15)
    batch = [synth_image(rng, 50, 70)]
    got = _run(batch, "HWC", axes=[0, 1], scale=64.0, shift=128.0, dtype=types.UINT8)
    ref = np.clip(np.rint(ON.normalize(batch[0], (0, 1), scale=64.0, shift=128.0)), 0, 255)
    assert got[0].dtype == np.uint8
    assert np.abs(got[0].astype(np.int32) - ref.astype(np.int32)).max() <= 1   # values exactly on .5 may round either way


def test_decoder_output_with_padded_rows(tmp_path):
    """decoders.image hands over row-padded images; normalize must see the pixels only."""
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(16)
    files = []
    for i, (h, w) in enumerate([(33, 47), (64, 50)]):     # 3*47 and 3*50 are not multiples of 16
        p = tmp_path / f"n{i}.jpg"
        p.write_bytes(encode_jpeg(synth_image(rng, h, w), 90))
        files.append(str(p))
    pipe = Pipeline(batch_size=2, num_threads=2, device_id=0)
    with pipe:
        enc, _ = fn.readers.file(files=files)
        pipe.set_outputs(fn.normalize(fn.decoders.image(enc, device="mixed"), axis_names="HW"))
    (out,) = pipe.run()
    for i, f in enumerate(files):
        img = O.jpeg_decode_rgb(open(f, "rb").read())
        _check([out[i].as_cpu()], [ON.normalize(img, (0, 1)).astype(np.float32)])


@pytest.mark.parametrize("channels", [1, 2, 3, 4])
def test_u8_images_of_one_to_four_channels_exact_means(channels):
    """The 16-bytes-per-lane path of the statistics kernels (u8, inner <= 4): ragged tails, several 65536-element chunks
    (a chunk start is not a multiple of 3), every channel count.  With stddev=1 the output is fl(x - mean) and the u8
    mean is an exact integer sum -> bit-exact against numpy's fp64 mean."""
    rng = np.random.default_rng(20 + channels)
    shapes = [(1, 1), (3, 5), (37, 53), (257, 259), (300, 437), (16, 4096)]
    batch = [rng.integers(0, 256, (h, w, channels), dtype=np.uint8) for (h, w) in shapes]
    got = _run(batch, "HWC", axes=[0, 1], stddev=1.0)
    for g, b in zip(got, batch):
        m = b.reshape(-1, channels).astype(np.float64).mean(axis=0).astype(np.float32)
        assert np.array_equal(g, b.astype(np.float32) - m)
    got = _run(batch, "HWC", axes=[0, 1])
    _check(got, [ON.normalize(b, (0, 1)).astype(np.float32) for b in batch], rtol=1e-5, atol=1e-5)
    got = _run(batch, "HWC")                      # one bin per sample: inner == 1, reduced = H * W * C
    _check(got, [ON.normalize(b, None).astype(np.float32) for b in batch], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("out_name", ["FLOAT16", "INT8", "UINT8"])
def test_narrow_outputs_and_row_statistics(out_name):
    """Vector stores of the apply pass for every output type; per-row statistics (outer = H, planes of odd length, so
    most row starts are unaligned and 4-element groups straddle rows)."""
    from dali_amd import types
    rng = np.random.default_rng(31)
    batch = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for (h, w) in [(37, 53), (5, 1), (64, 171), (130, 333)]]
    got = _run(batch, "HWC", axes=[1, 2], scale=40.0, shift=100.0 if out_name == "UINT8" else 0.0,
               dtype=getattr(types, out_name))
    for g, b in zip(got, batch):
        ref = ON.normalize(b, (1, 2), scale=40.0, shift=100.0 if out_name == "UINT8" else 0.0)
        if out_name == "FLOAT16":
            assert g.dtype == np.float16 and np.allclose(g.astype(np.float32), ref, rtol=2e-3, atol=2e-2)
        else:
            lo, hi = (0, 255) if out_name == "UINT8" else (-128, 127)
            assert np.abs(g.astype(np.int32) - np.clip(np.rint(ref), lo, hi).astype(np.int32)).max() <= 1


def test_non_adjacent_axes_are_rejected():
    with pytest.raises(RuntimeError, match="adjacent"):
        _run([np.zeros((4, 5, 3), np.float32)], "HWC", axes=[0, 2])
