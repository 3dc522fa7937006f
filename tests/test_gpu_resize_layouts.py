"""Boundary of the resizing operators:
 * `interp_type` / `min_filter` / `mag_filter` as TENSOR arguments, one value per sample (the reference's schema marks them
   tensor-argument capable, dali/operators/image/resize/resampling_attr.cc:25-37,76-121);
 * sequences and the other layouts (`.AllowSequences()`; random_resized_crop.cc:25-37, resize_base.cc:33-52): every layout
   in which W follows H - the dimensions in front of H are frames that share the sample's arguments, those behind W
   channels.  FHWC video, CHW / FCHW / CFHW planar data.
Each against the oracle, bit for bit: a frame of a sample must equal the oracle's resize of that image with the sample's
window and filters."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import synth_image

pytestmark = pytest.mark.gpu

FILTERS = {0: O.FILTER_NN, 1: O.FILTER_LINEAR, 2: O.FILTER_CUBIC, 3: O.FILTER_LANCZOS3, 4: O.FILTER_TRIANGULAR}


def test_per_sample_interpolation_types():
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(41)
    imgs = [synth_image(rng, h, w) for h, w in [(120, 160), (200, 150), (97, 131), (64, 48), (333, 500)]]
    interp = [np.array(v, np.int32) for v in (1, 2, 3, 4, 0)]          # DALIInterpType per sample
    mag = [np.array(v, np.int32) for v in (2, 1, 1, 3, 1)]
    pipe = Pipeline(batch_size=5, num_threads=2, device_id=0, prefetch_queue_depth=1)
    with pipe:
        x = fn.external_source(name="images", layout="HWC")
        it = fn.external_source(name="interp")
        mg = fn.external_source(name="mag")
        a = fn.resize(x.gpu(), size=[90, 110], interp_type=it)                 # both filters from the tensor
        b = fn.resize(x.gpu(), size=[90, 110], min_filter=it, mag_filter=mg)   # ... and each from its own
        c = fn.resize(x.gpu(), size=[300, 400], interp_type=it, mag_filter=mg)  # mag_filter wins over interp_type
        pipe.set_outputs(a, b, c)
    pipe.build()
    pipe.feed_input("images", imgs, layout="HWC")
    pipe.feed_input("interp", interp)
    pipe.feed_input("mag", mag)
    a, b, c = pipe.run()
    for i, im in enumerate(imgs):
        f, m = FILTERS[int(interp[i])], FILTERS[int(mag[i])]
        # (antialias, the default: LINEAR as a minification filter is the triangular one - resampling_attr.cc:96-107)
        fmin = O.FILTER_TRIANGULAR if f == O.FILTER_LINEAR else f
        assert np.array_equal(a[i].as_cpu(), O.resample_u8(im, (90, 110), min_filter=fmin, mag_filter=f)), ("a", i)
        assert np.array_equal(b[i].as_cpu(), O.resample_u8(im, (90, 110), min_filter=fmin, mag_filter=m)), ("b", i)
        assert np.array_equal(c[i].as_cpu(), O.resample_u8(im, (300, 400), min_filter=fmin, mag_filter=m)), ("c", i)


def _frames(sample, layout):
    """The H x W x C images of a sample in `layout`, in the order of the flattened leading dimensions."""
    hi = layout.index("H")
    lead = int(np.prod(sample.shape[:hi], dtype=np.int64))
    h, w = sample.shape[hi], sample.shape[hi + 1]
    return sample.reshape(lead, h, w, -1)


@pytest.mark.parametrize("layout,shape", [("FHWC", (5, 70, 90, 3)), ("CHW", (3, 70, 90)), ("FCHW", (4, 3, 50, 64)),
                                          ("CFHW", (2, 3, 50, 64)), ("HWC", (70, 90, 3)), ("FHWC", (3, 41, 37, 1))])
def test_resize_of_sequences_and_planar_layouts(layout, shape):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(42)
    n = 3
    data = [rng.integers(0, 256, tuple(np.array(shape) + (0 if k == 0 else 0)), dtype=np.uint8) for k in range(n)]
    # (samples of different spatial sizes: H and W vary, the other dimensions are the layout's)
    hi = layout.index("H")
    for k in range(1, n):
        s = list(shape)
        s[hi] += 7 * k
        s[hi + 1] -= 5 * k
        data[k] = rng.integers(0, 256, tuple(s), dtype=np.uint8)
    pipe = Pipeline(batch_size=n, num_threads=2, device_id=0, prefetch_queue_depth=1)
    with pipe:
        x = fn.external_source(name="x", layout=layout)
        pipe.set_outputs(fn.resize(x.gpu(), size=[33, 47]), fn.resize(x.gpu(), resize_shorter=40.0, interp_type=2))
    pipe.build()
    pipe.feed_input("x", data, layout=layout)
    fixed, shorter = pipe.run()
    for i, smp in enumerate(data):
        got = np.asarray(fixed[i].as_cpu())
        want_shape = list(smp.shape)
        want_shape[hi], want_shape[hi + 1] = 33, 47
        assert list(got.shape) == want_shape, (got.shape, want_shape)
        for f, (g, src) in enumerate(zip(_frames(got, layout), _frames(smp, layout))):
            assert np.array_equal(g, O.resample_u8(src, (33, 47), min_filter=O.FILTER_TRIANGULAR)), (i, f)
        got2 = np.asarray(shorter[i].as_cpu())
        out_hw, roi = O.resize_params(smp.shape[hi:hi + 2], size=(40, 40), mode="not_smaller")
        for f, (g, src) in enumerate(zip(_frames(got2, layout), _frames(smp, layout))):
            assert np.array_equal(g, O.resample_u8(src, out_hw, roi=roi, min_filter=O.FILTER_CUBIC, mag_filter=O.FILTER_CUBIC)), (i, f)


def test_random_resized_crop_of_video_uses_one_window_per_sequence():
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(43)
    vids = [rng.integers(0, 256, (f, h, w, 3), dtype=np.uint8) for f, h, w in [(4, 120, 160), (2, 90, 200), (6, 64, 64)]]
    pipe = Pipeline(batch_size=3, num_threads=2, device_id=0, prefetch_queue_depth=1)
    with pipe:
        x = fn.external_source(name="x", layout="FHWC")
        pipe.set_outputs(fn.random_resized_crop(x.gpu(), size=[48, 56], seed=321))
    pipe.build()
    for it in range(2):
        pipe.feed_input("x", vids, layout="FHWC")
        (out,) = pipe.run()
        anchors, crops = O.rrc_batch(321, it, [v.shape[1:3] for v in vids])
        for i, v in enumerate(vids):
            got = np.asarray(out[i].as_cpu())
            assert got.shape == (v.shape[0], 48, 56, 3)
            (y0, x0), (h, w) = anchors[i], crops[i]
            for f in range(v.shape[0]):
                ref = O.resample_u8(v[f], (48, 56), roi=(y0, x0, y0 + h, x0 + w), min_filter=O.FILTER_TRIANGULAR)
                assert np.array_equal(got[f], ref), (it, i, f)


def test_layouts_without_adjacent_h_and_w_are_refused():
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    pipe = Pipeline(batch_size=1, num_threads=1, device_id=0, prefetch_queue_depth=1)
    with pipe:
        x = fn.external_source(name="x", layout="HCW")
        pipe.set_outputs(fn.resize(x.gpu(), size=[8, 8]))
    pipe.build()
    pipe.feed_input("x", [np.zeros((10, 3, 12), np.uint8)], layout="HCW")
    with pytest.raises(RuntimeError, match="W follows H"):
        pipe.run()
