"""GPU parity: fused separable resampling (+CMN epilogue) vs the CPU oracle.

Tolerance (stated, fp32 interpolation): every u8 output within 1 LSB of the oracle and at most
0.1 % of the elements different at all.  The kernel follows the CPU backend's arithmetic order
(pre-normalised coefficients, separate mul/add, reference pass order, SSE2-body/tail rounding split),
so in practice the result is expected to be bit-identical; the assertions print the observed
mismatch count."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.util import synth_image

pytestmark = pytest.mark.gpu


def _to_dev(img, pitch_align=1):
    h, w, c = img.shape
    pitch = (w * c + pitch_align - 1) // pitch_align * pitch_align
    buf = torch.zeros(h * pitch + 64, dtype=torch.uint8, device="cuda")
    view = torch.as_strided(buf, (h, w, c), (pitch, c, 1), 0)
    view.copy_(torch.from_numpy(img))
    return view


def _check(got, ref, what):
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    nbad = int((d > 0).sum())
    print(f"{what}: max diff {d.max()}, mismatches {nbad}/{d.size}")
    assert d.max() <= 1, what
    assert nbad <= max(1, d.size // 1000), what


@pytest.mark.parametrize("pitch_align", [1, 4, 64])
def test_rrc_like_rois_match_oracle(pitch_align):
    from dali_amd import backend as B
    rng = np.random.default_rng(5)
    sizes = [(375, 500), (500, 375), (480, 640), (333, 500), (256, 384), (97, 131), (768, 1024)]
    imgs, rois = [], []
    for i in range(21):
        h, w = sizes[i % len(sizes)]
        imgs.append(synth_image(rng, h, w))
    anchors, crops = O.rrc_batch(99, 0, [im.shape[:2] for im in imgs])
    for i in range(len(imgs)):
        rois.append((anchors[i][0], anchors[i][1], anchors[i][0] + crops[i][0], anchors[i][1] + crops[i][1]))
    dev = [_to_dev(im, pitch_align) for im in imgs]
    out = B.resample_batch(dev, (224, 224), rois=rois)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    orders = []
    for i, im in enumerate(imgs):
        ref, info = O.resample_u8(im, (224, 224), roi=rois[i], return_info=True)
        orders.append(int(info[0]))
        _check(out[i], ref, f"sample {i} shape {im.shape} roi {rois[i]} first_axis {info[0]}")
    assert 0 in orders and 1 in orders, "both pass orders must be exercised"


def test_whole_image_up_and_down_scaling():
    from dali_amd import backend as B
    rng = np.random.default_rng(6)
    cases = [((64, 48), (224, 224)), ((300, 300), (93, 479)), ((479, 93), (93, 93)), ((1000, 40), (50, 50)),
             ((33, 777), (64, 64)), ((16, 16), (16, 16)), ((1, 1), (7, 5)), ((2, 300), (300, 2))]
    for (h, w), osz in cases:
        for c in (1, 3, 4):
            im = synth_image(rng, h, w, 3)
            im = im[:, :, :1] if c == 1 else (np.concatenate([im, im[:, :, :1]], 2) if c == 4 else im)
            im = np.ascontiguousarray(im)
            out = B.resample_batch([_to_dev(im)], osz).cpu().numpy()[0]
            ref = O.resample_u8(im, osz)
            _check(out, ref, f"{(h, w, c)} -> {osz}")


def test_linear_no_antialias_and_flipped_roi():
    from dali_amd import backend as B
    from dali_amd import _capi as capi
    rng = np.random.default_rng(8)
    im = synth_image(rng, 200, 300)
    dev = _to_dev(im, 4)
    out = B.resample_batch([dev], (100, 120), antialias=False).cpu().numpy()[0]
    _check(out, O.resample_u8(im, (100, 120), antialias=False), "no antialias")
    roi = (150.0, 280.0, 20.0, 10.0)  # flipped in both axes
    out = B.resample_batch([dev], (64, 96), rois=[roi]).cpu().numpy()[0]
    _check(out, O.resample_u8(im, (64, 96), roi=roi), "flipped roi")
    roi = (10.5, 20.25, 150.75, 220.5)  # fractional
    out = B.resample_batch([dev], (224, 224), rois=[roi]).cpu().numpy()[0]
    _check(out, O.resample_u8(im, (224, 224), roi=roi), "fractional roi")


def test_extreme_downscale_uses_small_tiles():
    from dali_amd import backend as B
    rng = np.random.default_rng(9)
    im = synth_image(rng, 1500, 2000)
    out = B.resample_batch([_to_dev(im, 4)], (32, 32)).cpu().numpy()[0]
    _check(out, O.resample_u8(im, (32, 32)), "2000x1500 -> 32x32")


@pytest.mark.parametrize("dtype", ["float16", "float32"])
@pytest.mark.parametrize("layout", ["CHW", "HWC"])
def test_fused_rrc_cmn_matches_oracle_composition(dtype, layout):
    """RRC -> CMN fused on the GPU == oracle RRC (u8) followed by oracle CMN, bit for bit wherever the
    u8 intermediate agrees."""
    from dali_amd import backend as B
    from dali_amd import _capi as capi
    rng = np.random.default_rng(10)
    imgs = [synth_image(rng, h, w) for (h, w) in [(375, 500), (500, 375), (480, 640), (256, 384)] * 2]
    anchors, crops = O.rrc_batch(4321, 3, [im.shape[:2] for im in imgs])
    rois = [(a[0], a[1], a[0] + c[0], a[1] + c[1]) for a, c in zip(anchors, crops)]
    mirror = O.coin_flip_batch(17, 3, len(imgs), 0.5)
    mean, inv = O.cmn_norm_args([0.485 * 255, 0.456 * 255, 0.406 * 255], [0.229 * 255, 0.224 * 255, 0.225 * 255])
    dt = capi.FLOAT16 if dtype == "float16" else capi.FLOAT
    lay = capi.LAYOUT_CHW if layout == "CHW" else capi.LAYOUT_HWC
    out = B.resample_batch([_to_dev(im, 16) for im in imgs], (224, 224), rois=rois, out_dtype=dt, out_layout=lay,
                           mean=mean, inv_std=inv, mirror=mirror)
    u8 = B.resample_batch([_to_dev(im, 16) for im in imgs], (224, 224), rois=rois).cpu().numpy()
    out = out.cpu().numpy()
    for i, im in enumerate(imgs):
        # the GPU's own u8 result, normalised by the oracle, must equal the fused output exactly
        ref = O.cmn_u8(u8[i], (0, 0), (224, 224), mirror=bool(mirror[i]), mean=mean, inv_std=inv, layout=layout,
                       dtype=O.F16 if dtype == "float16" else O.F32)
        assert np.array_equal(out[i].view(np.uint16 if dtype == "float16" else np.uint32),
                              ref.view(np.uint16 if dtype == "float16" else np.uint32)), f"sample {i}"
        # and the u8 intermediate obeys the resampling tolerance
        _check(u8[i], O.resample_u8(im, (224, 224), roi=rois[i]), f"u8 sample {i}")


@pytest.mark.parametrize("out_hw", [(37, 50), (45, 51), (16, 2), (100, 98), (33, 34), (224, 226)])
def test_fused_fp16_chw_ragged_sizes_both_pass_orders(out_hw):
    """The fp16 CHW epilogue takes two pixels per thread when the width is even (odd widths, 1-pixel tiles and other
    layouts keep the one-pixel path): partial tiles, both pass orders (wide and tall sources), mirrored or not, against
    the u8 kernel path + oracle CMN bit for bit and against the oracle's resampling."""
    from dali_amd import backend as B
    from dali_amd import _capi as capi
    rng = np.random.default_rng(12)
    imgs = [synth_image(rng, h, w) for (h, w) in [(90, 400), (400, 90), (130, 170), (64, 64), (301, 203), (203, 301)]]
    rois = [(3.0, 5.0, h - 2.0, w - 4.0) if i % 2 else None for i, (h, w) in enumerate(im.shape[:2] for im in imgs)]
    mirror = np.array([0, 1, 1, 0, 1, 0], np.int32)
    mean, inv = O.cmn_norm_args([0.485 * 255, 0.456 * 255, 0.406 * 255], [0.229 * 255, 0.224 * 255, 0.225 * 255])
    out, descs, _, _ = B.resample_batch([_to_dev(im, 16) for im in imgs], out_hw, rois=rois, out_dtype=capi.FLOAT16,
                                        out_layout=capi.LAYOUT_CHW, mean=mean, inv_std=inv, mirror=mirror, return_descs=True)
    assert set(descs["first_axis"].tolist()) == {0, 1}, "both pass orders must be exercised"
    u8 = B.resample_batch([_to_dev(im, 16) for im in imgs], out_hw, rois=rois).cpu().numpy()
    out = out.cpu().numpy()
    for i, im in enumerate(imgs):
        ref = O.cmn_u8(u8[i], (0, 0), out_hw, mirror=bool(mirror[i]), mean=mean, inv_std=inv, layout="CHW", dtype=O.F16)
        assert np.array_equal(out[i].view(np.uint16), ref.view(np.uint16)), f"sample {i}"
        _check(u8[i], O.resample_u8(im, out_hw, roi=rois[i]), f"u8 sample {i}")


def test_checkerboard_vs_onnx_reference_on_the_gpu():
    """The reference's golden pin (test_resize.py:919-1029, atol 1) through the HIP kernel: 22 x 22 checkerboard ->
    17 x 13, antialiased linear (the kernel's triangular filter), against the ONNX reference restated in
    tests/onnx_resize_ref.py - independent of the oracle."""
    from dali_amd import backend as B
    from tests import onnx_resize_ref as R
    board = R.checkerboard_22_22()
    ref = R.interpolate_nd(board, R.linear_coeffs_antialias, (17, 13))
    img = np.repeat(board[:, :, None], 3, axis=2)
    out = B.resample_batch([_to_dev(img)], (17, 13))
    torch.cuda.synchronize()
    got = out[0].cpu().numpy().astype(np.float64)
    for c in range(3):
        assert np.abs(got[:, :, c] - ref).max() <= 1.0, c
    assert np.array_equal(out[0].cpu().numpy(), O.resample_u8(img, (17, 13)))


@pytest.mark.parametrize("filt", ["NN", "CUBIC", "LANCZOS3", "GAUSSIAN"])
def test_other_filters_match_oracle(filt):
    """Nearest neighbour and the reference's tabulated windows (cubic, Lanczos3, Gaussian): up- and down-scaling with
    and without antialiasing, regions of interest, both pass orders - bit-identical to the oracle."""
    from dali_amd import backend as B
    from dali_amd import _capi as capi
    kf, of = getattr(capi, "INTERP_" + filt), getattr(O, "FILTER_" + filt)
    rng = np.random.default_rng(12)
    cases = [((120, 160), (224, 224), None, True), ((300, 400), (93, 131), None, True), ((300, 400), (93, 131), None, False),
             ((200, 90), (64, 200), (10.5, 5.25, 180.0, 80.5), True), ((97, 131), (97, 131), None, True),
             ((64, 64), (7, 300), None, True), ((40, 300), (120, 60), (35.0, 290.0, 2.0, 11.0), True)]
    for (h, w), osz, roi, aa in cases:
        for c in (1, 3):
            im = np.ascontiguousarray(synth_image(rng, h, w, 3)[:, :, :c])
            out = B.resample_batch([_to_dev(im, 4)], osz, rois=None if roi is None else [roi], interp_min=kf, interp_mag=kf,
                                   antialias=aa).cpu().numpy()[0]
            ref = O.resample_u8(im, osz, roi=roi, min_filter=of, mag_filter=of, antialias=aa)
            assert np.array_equal(out, ref), (filt, (h, w, c), osz, roi, aa, int(np.abs(out.astype(int) - ref).max()))


def test_nearest_on_one_axis_only_is_refused():
    from dali_amd import backend as B
    from dali_amd import _capi as capi
    im = synth_image(np.random.default_rng(1), 50, 80)
    with pytest.raises(capi.DaliAmdError, match="one axis only"):
        B.resample_batch([_to_dev(im)], (100, 20), interp_min=capi.INTERP_NN, interp_mag=capi.INTERP_LINEAR)


@pytest.mark.parametrize("np_t", [np.int16, np.uint16, np.float32, np.uint8])
def test_other_element_types_match_oracle(np_t):
    """i16 / u16 / f32 images (output = input type) and the unrounded float result of u8 / i16 images: the two-launch
    path of the kernel library against the oracle, bit for bit, in both pass orders, with a region of interest."""
    from dali_amd import backend as B
    rng = np.random.default_rng(14)
    conv = {np.int16: lambda a: np.clip(a * 9000, -32768, 32767), np.uint16: lambda a: np.clip(a * 9000 + 30000, 0, 65535),
            np.float32: lambda a: a * 3, np.uint8: lambda a: np.clip(a * 60 + 128, 0, 255)}[np_t]
    orders = set()
    for (h, w, c), osz, roi in [((60, 90, 3), (40, 50), None), ((33, 47, 1), (150, 70), None), ((120, 80, 3), (24, 64), None),
                                ((200, 300, 3), (64, 64), (20.5, 30.0, 180.0, 250.25)), ((90, 400, 3), (224, 32), None)]:
        im = conv(rng.normal(0, 1, (h, w, c))).astype(np_t)
        dev = torch.from_numpy(im).cuda()
        for unrounded in ((True,) if np_t == np.uint8 else (False, True)):
            if unrounded and np_t == np.float32:
                continue
            out = B.resample_batch([dev], osz, rois=None if roi is None else [roi], out_dtype=None, unrounded=unrounded)
            got = out.cpu().numpy()[0]
            ref = O.resample_typed(im, osz, out_type=O.T_F32 if unrounded else None, roi=roi)
            assert got.dtype == ref.dtype and got.shape == ref.shape
            assert np.array_equal(got, ref), (np_t, (h, w, c), osz, unrounded, np.abs(got.astype(np.float64) - ref).max())
        _, info = O.resample_u8(np.zeros((h, w, c), np.uint8), osz, roi=roi, return_info=True)
        orders.add(int(info[0]))
    assert orders == {0, 1}, "both pass orders must be exercised"


def test_resize_operator_other_types_on_the_gpu():
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(15)
    imgs = [(rng.normal(0, 1, (70, 50, 3)) * 8000).astype(np.int16), (rng.normal(0, 1, (31, 64, 3)) * 8000).astype(np.int16)]
    for dtype in (None, types.FLOAT):
        pipe = Pipeline(batch_size=2, num_threads=1, device_id=0, prefetch_queue_depth=1)
        with pipe:
            x = fn.external_source(name="x", layout="HWC")
            kw = {} if dtype is None else {"dtype": dtype}
            pipe.set_outputs(fn.resize(x.gpu(), resize_shorter=40, **kw))
        pipe.feed_input("x", imgs, layout="HWC")
        (out,) = pipe.run()
        for i, im in enumerate(imgs):
            out_hw, roi = O.resize_params(im.shape[:2], size=(40, 40), mode="not_smaller")
            ref = O.resample_typed(im, out_hw, out_type=O.T_F32 if dtype is not None else None, roi=roi)
            got = out[i].as_cpu()
            assert got.dtype == ref.dtype and np.array_equal(got, ref), (dtype, i)


def _tie_rich(rng, h, w):
    """Pixels whose 2x antialiased down-scale (taps 1/8 3/8 3/8 1/8 per axis: results are multiples of 1/64) lands on
    exact .5 ties in about one output out of 64 - what tells the SIMD store's half-to-even from the scalar tail's
    half-away rounding."""
    return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)


@pytest.mark.parametrize("case", [
    ((512, 682), (256, 341), None),                  # the ImageNet validation shape, exact 2x: ties everywhere
    ((512, 1500), (256, 750), None),                 # three times the old 256-column mask
    ((520, 700), (256, 341), (4.0, 9.0, 516.0, 691.0)),        # ROI: clamped border regions on both sides
    ((520, 700), (256, 341), (4.0, 691.0, 516.0, 9.0)),        # ... flipped
    ((375, 500), (256, 341), None), ((480, 640), (256, 341), None),   # resize_shorter=256 of the usual ImageNet sizes
    ((600, 1100), (300, 550), None),
])
def test_wide_outputs_round_in_the_reference_regions(case):
    """Outputs wider than 256 columns with an H-last pass: columns in whole 16-lane groups of each region round half to
    even, the tail columns of the regions half away (simd.h:53-56 vs convert.h:306-321) - bit for bit like the oracle,
    and NOT like "half to even everywhere" (round 2's approximation for wide outputs) when the image has ties there."""
    from dali_amd import backend as B
    (h, w), osz, roi = case
    rng = np.random.default_rng(h * 7 + w)
    im = _tie_rich(rng, h, w)
    rois = None if roi is None else [roi]
    out = B.resample_batch([_to_dev(im, 16)], osz, rois=rois).cpu().numpy()[0]
    ref, info = O.resample_u8(im, osz, roi=roi, return_info=True)
    assert int(info[0]) == 1, "the case must take the vertical pass first (H-last)"
    assert np.array_equal(out, ref), f"{np.argwhere(out != ref)[:5]}"
    if (h, w) == (512, 682) or (h, w) == (512, 1500):
        all_even = O.resample_u8(im, osz, roi=roi, round_mode=2)
        assert (all_even != ref).any(), "the case holds no tie in a tail column: it would not have caught the old mask"
        assert (out != all_even).any()


def test_wide_fused_fp16_output_rounds_in_the_reference_regions():
    """Same through the fused CMN epilogue (pixel pairs + look-up table): fn.resize -> crop_mirror_normalize shapes."""
    from dali_amd import backend as B
    from dali_amd import _capi as capi
    rng = np.random.default_rng(77)
    im = _tie_rich(rng, 512, 684)
    mean, inv = O.cmn_norm_args([0.485 * 255, 0.456 * 255, 0.406 * 255], [0.229 * 255, 0.224 * 255, 0.225 * 255])
    for mirror in (0, 1):
        out = B.resample_batch([_to_dev(im, 16)], (256, 342), out_dtype=capi.FLOAT16, out_layout=capi.LAYOUT_CHW, mean=mean,
                               inv_std=inv, mirror=np.array([mirror], np.uint8)).cpu().numpy()[0]
        u8 = O.resample_u8(im, (256, 342))
        ref = O.cmn_u8(u8, (0, 0), (256, 342), mirror=bool(mirror), mean=mean, inv_std=inv, dtype=O.F16)
        assert np.array_equal(out.view(np.uint16), ref.view(np.uint16)), mirror


def _kernel_launches():
    import ctypes as C
    from dali_amd import _capi as capi
    lib = capi.kernels()
    need = lib.daliamdKernelTimingReport(None, 0)
    buf = C.create_string_buffer(need + 1)
    lib.daliamdKernelTimingReport(buf, need + 1)
    return {ln.split("\t")[0]: int(ln.split("\t")[1]) for ln in buf.value.decode().splitlines()}


@pytest.mark.parametrize("dtype", ["uint8", "float16"])
def test_twelve_megapixel_crops_take_the_two_launch_path_in_a_mixed_batch(dtype):
    """Round 6: a 12-megapixel photograph among ImageNet-sized images (the `large` data set variant of bench.py).  Its
    crop windows shrink 9-13 x into 224 x 224: the tile kernel would serve them with 4 x 4-pixel tiles (1.2 ms per batch for
    five such images); SetupOne routes them to the two-launch path with the fused epilogue (generic = 2).  Same bits as the
    oracle - both pass orders, mirrored, normalised fp16 CHW - next to ordinary samples on the tile path in one Run."""
    from dali_amd import backend as B
    from dali_amd import _capi as capi
    rng = np.random.default_rng(31)
    base = synth_image(rng, 750, 1000)
    big = np.ascontiguousarray(np.kron(base, np.ones((4, 4, 1), np.uint8)))            # 3000 x 4000
    big = (big.astype(np.int16) + rng.integers(-20, 21, big.shape, dtype=np.int16)).clip(0, 255).astype(np.uint8)
    imgs = [big, synth_image(rng, 375, 500), big, synth_image(rng, 480, 640), big]
    rois = [(100.0, 200.0, 2900.0, 3300.0),      # wide: 2800 x 3100
            (10.0, 20.0, 300.0, 420.0),
            (1000.5, 50.25, 1700.0, 3950.75),    # short, wide and fractional (700 x 3900): the horizontal pass goes first
            (0.0, 0.0, 480.0, 640.0),
            (2999.0, 3999.0, 0.0, 0.0)]          # the whole image, flipped in both axes
    mirror = np.array([1, 0, 0, 1, 1], np.int32)
    mean, inv = O.cmn_norm_args([0.485 * 255, 0.456 * 255, 0.406 * 255], [0.229 * 255, 0.224 * 255, 0.225 * 255])
    lib = capi.kernels()
    _kernel_launches()
    lib.daliamdKernelTimingEnable(1)
    dev = [_to_dev(im, 16) for im in imgs]
    if dtype == "uint8":
        out = B.resample_batch(dev, (224, 224), rois=rois).cpu().numpy()
    else:
        out = B.resample_batch(dev, (224, 224), rois=rois, out_dtype=capi.FLOAT16, out_layout=capi.LAYOUT_CHW,
                               mean=mean, inv_std=inv, mirror=mirror).cpu().numpy()
    torch.cuda.synchronize()
    lib.daliamdKernelTimingEnable(0)
    launched = _kernel_launches()
    assert launched.get("ResampleGenericKernel", 0) >= 1, launched
    import os
    if int(os.environ.get("DALI_AMD_RESAMPLE_TWO_PASS_AREA", "64")) <= 512:     # (the suite is also run with everything forced there)
        assert launched.get("ResampleKernel", 0) >= 1, launched
    orders, bad = [], []
    for i, im in enumerate(imgs):
        ref, info = O.resample_u8(im, (224, 224), roi=rois[i], return_info=True)
        orders.append(int(info[0]))
        if dtype == "uint8":
            if not np.array_equal(out[i], ref):
                d = np.abs(out[i].astype(int) - ref.astype(int))
                bad.append((i, int((d > 0).sum()), int(d.max()), np.argwhere(d > 0)[:4].tolist()))
        else:
            want = O.cmn_u8(ref, (0, 0), (224, 224), mirror=bool(mirror[i]), mean=mean, inv_std=inv, dtype=O.F16)
            if not np.array_equal(out[i].view(np.uint16), want.view(np.uint16)):
                bad.append((i, int((out[i].view(np.uint16) != want.view(np.uint16)).sum())))
    assert not bad, f"(sample, differing elements, ...): {bad}; pass orders {orders}"
    assert {orders[0], orders[2]} == {0, 1}, f"both pass orders on the large samples: {orders}"
