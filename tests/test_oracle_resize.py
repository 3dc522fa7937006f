"""fn.resize size arithmetic of the oracle, pinned by the worked examples in the reference's documentation strings
(dali/operators/image/resize/resize_attr_base.cc:33-41 and :81-85) and by the rules stated in resize_attr.cc:28-58."""
import numpy as np
import pytest

from oracle import oracle as O


def test_documented_examples():
    # "a 1280x720, with a desired output size of 640x480, actually produces a 640x360 output" (not_larger; W x H order)
    assert O.resize_params((720, 1280), (480, 640), "not_larger")[0] == (360, 640)
    # "a 640x480 image with a desired output size of 1920x1080, actually produces a 1920x1440 output" (not_smaller)
    assert O.resize_params((480, 640), (1080, 1920), "not_smaller")[0] == (1440, 1920)
    # "mode=not_smaller, size=800, max_size=1400 an image of size 1200x600 would be resized to 1400x700"
    assert O.resize_params((600, 1200), (800, 800), "not_smaller", max_size=1400)[0] == (700, 1400)


def test_missing_extent_keeps_aspect_ratio_in_default_mode_and_size_in_stretch_mode():
    assert O.resize_params((300, 400), (0, 200), "default")[0] == (150, 200)     # resize_x only
    assert O.resize_params((300, 400), (150, 0), "default")[0] == (150, 200)     # resize_y only
    assert O.resize_params((300, 400), (0, 200), "stretch")[0] == (300, 200)
    assert O.resize_params((300, 400), (0, 0), "default")[0] == (300, 400)       # nothing given: keep the size


def test_resize_shorter_and_longer():
    # resize_shorter = same size for all dimensions + not_smaller; resize_longer = ... + not_larger
    assert O.resize_params((375, 500), (256, 256), "not_smaller")[0] == (256, 341)
    assert O.resize_params((500, 375), (256, 256), "not_smaller")[0] == (341, 256)
    assert O.resize_params((375, 500), (256, 256), "not_larger")[0] == (192, 256)
    assert O.resize_params((375, 500), (256, 256), "not_smaller", max_size=300)[0] == (225, 300)


def test_subpixel_scale_adjusts_the_region_around_its_centre():
    # 500 -> 341.33: the output is 341 pixels and the source region shrinks by 341/341.33 around its centre
    (oh, ow), (y0, x0, y1, x1) = O.resize_params((375, 500), (256, 256), "not_smaller")
    assert (oh, ow) == (256, 341) and (y0, y1) == (0.0, 375.0)
    k = 341 / (500 * np.float32(256 / 375))
    assert x0 == pytest.approx(250 - 250 * k, abs=1e-4) and x1 == pytest.approx(250 + 250 * k, abs=1e-4)
    # without it the whole image maps to the rounded size
    assert O.resize_params((375, 500), (256, 256), "not_smaller", subpixel_scale=False)[1] == (0.0, 0.0, 375.0, 500.0)


def test_region_of_interest_absolute_relative_and_degenerate():
    assert O.resize_params((100, 200), (50, 50), roi=(10, 20, 60, 120))[1] == (10.0, 20.0, 60.0, 120.0)
    assert O.resize_params((100, 200), (50, 50), roi=(0.1, 0.1, 0.6, 0.6), roi_relative=True)[1] == \
        pytest.approx((10.0, 20.0, 60.0, 120.0))
    # a region flipped in x: the requested size changes sign internally, lo/hi come out swapped (mirrored sampling)
    (oh, ow), (y0, x0, y1, x1) = O.resize_params((100, 200), (50, 50), roi=(10, 120, 60, 20))
    assert (oh, ow) == (50, 50) and x0 > x1 and (y0, y1) == (10.0, 60.0)
    # a degenerate region is widened to 1e-3 pixels instead of failing
    (oh, ow), (y0, x0, y1, x1) = O.resize_params((100, 200), (8, 8), roi=(10, 50, 60, 50))
    assert x1 - x0 == pytest.approx(1e-3, rel=1e-2)


def test_output_is_at_least_one_pixel():
    assert O.resize_params((1000, 10), (10, 0), "default")[0] == (10, 1)


@pytest.mark.parametrize("mirror", [0, 1, 2, 3])
def test_resize_crop_mirror_equals_resize_then_crop_then_flip(mirror):
    """The reference's own check of the fused operator (test_resize_crop_mirror.py:27-84): one resampling of the
    back-projected window == resize -> crop -> flip, max difference 1."""
    from tests.util import synth_image
    rng = np.random.default_rng(11 + mirror)
    img = synth_image(rng, 150, 210)
    for size, mode, roi, crop, pos in [((96, 96), "not_smaller", None, (80, 72), (0.3, 0.9)),
                                       ((100, 130), "stretch", (0.3, 0.8, 0.9, 0.1), (70, 90), (0.4, 0.2)),
                                       ((120, 0), "default", (0.7, 0.2, 0.1, 0.8), (64, 64), (1.0, 0.0))]:
        kw = dict(size=size, mode=mode, roi=roi, roi_relative=roi is not None)
        out_hw, win = O.resize_crop_mirror_params(img.shape[:2], crop=crop, crop_pos=pos, mirror=mirror, **kw)
        fused = O.resample_u8(img, out_hw, roi=win)
        rs_hw, rs_roi = O.resize_params(img.shape[:2], **kw)
        resized = O.resample_u8(img, rs_hw, roi=rs_roi)
        ay, ax = O.crop_anchor(pos[0], crop[0], rs_hw[0]), O.crop_anchor(pos[1], crop[1], rs_hw[1])
        seq = resized[ay:ay + crop[0], ax:ax + crop[1]]
        if mirror & 1:
            seq = seq[:, ::-1]
        if mirror & 2:
            seq = seq[::-1]
        assert fused.shape == seq.shape == (crop[0], crop[1], 3)
        d = np.abs(fused.astype(int) - seq)
        assert d.max() <= 1 and d.mean() < 1e-3 * 255, (size, d.max(), d.mean())
