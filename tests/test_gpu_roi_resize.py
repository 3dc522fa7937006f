"""GPU end-to-end: region-of-interest decoders (decoders.image_crop / image_random_crop) and fn.resize through the
C++ pipeline, checked against the oracle compositions "decode, then crop" and "ResizeAttr arithmetic + resampling"."""
import numpy as np
import pytest
from PIL import Image, ImageOps

from oracle import oracle as O
from tests.util import encode_jpeg, synth_image

pytestmark = pytest.mark.gpu

MEAN = [0.485 * 255, 0.456 * 255, 0.406 * 255]
STD = [0.229 * 255, 0.224 * 255, 0.225 * 255]


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    root = tmp_path_factory.mktemp("roi")
    rng = np.random.default_rng(77)
    out = []
    specs = [((120, 160), dict(subsampling="4:2:0")), ((200, 150), dict(subsampling="4:4:4")),
             ((97, 131), dict(subsampling="4:2:2")), ((240, 320), dict(subsampling="4:2:0", progressive=True)),
             ((64, 48), dict(subsampling="4:1:1")), ((333, 500), dict(subsampling="4:2:0", restart_marker_blocks=7)),
             ((180, 180), dict(subsampling="4:2:0", optimize=True)), ((75, 211), dict(subsampling="4:2:0"))]
    for i, (hw, kw) in enumerate(specs):
        p = root / f"img{i}.jpg"
        p.write_bytes(encode_jpeg(synth_image(rng, *hw), 85, **kw))
        out.append(str(p))
    return out


def _decoded(files):
    return [O.jpeg_decode_rgb(open(f, "rb").read()) for f in files]


def test_image_random_crop_equals_decode_then_crop(files):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    bs = len(files)
    pipe = Pipeline(batch_size=bs, num_threads=3, device_id=0, prefetch_queue_depth=2)
    with pipe:
        enc, _ = fn.readers.file(files=files)
        pipe.set_outputs(fn.decoders.image_random_crop(enc, device="mixed", seed=1234, random_area=[0.1, 0.9]))
    ref = _decoded(files)
    for it in range(4):
        (out,) = pipe.run()
        assert "jpeg_huffman" in pipe.executed_kernels()
        anchors, crops = O.rrc_batch(1234, it, [r.shape[:2] for r in ref], area=(0.1, 0.9))
        for i in range(bs):
            (y0, x0), (h, w) = anchors[i], crops[i]
            got = out[i].as_cpu()
            assert got.shape == (h, w, 3), (it, i)
            assert np.array_equal(got, ref[i][y0:y0 + h, x0:x0 + w]), (it, i)


def test_image_crop_fixed_window_and_per_sample_anchor(files):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    bs = len(files)
    pos_x = np.linspace(0, 1, bs).astype(np.float32)
    pipe = Pipeline(batch_size=bs, num_threads=2, device_id=0)
    with pipe:
        enc, _ = fn.readers.file(files=files)
        px = fn.external_source(source=lambda: [np.array(v, np.float32) for v in pos_x], batch=True)
        pipe.set_outputs(fn.decoders.image_crop(enc, device="mixed", crop=(40, 33), crop_pos_x=px, crop_pos_y=0.3))
    (out,) = pipe.run()
    for i, r in enumerate(_decoded(files)):
        y0 = O.crop_anchor(0.3, 40, r.shape[0])
        x0 = O.crop_anchor(float(pos_x[i]), 33, r.shape[1])
        assert np.array_equal(out[i].as_cpu(), r[y0:y0 + 40, x0:x0 + 33]), i


def test_image_crop_with_exif_orientation(tmp_path):
    """The window is a window of the UPRIGHT image (image_decoder.h:676-684)."""
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(5)
    paths = []
    for o in range(1, 9):
        img = Image.fromarray(synth_image(rng, 72 + 8 * o, 100))
        exif = Image.Exif()
        exif[0x0112] = o
        p = tmp_path / f"o{o}.jpg"
        img.save(p, "JPEG", quality=90, exif=exif)
        paths.append(str(p))
    pipe = Pipeline(batch_size=8, num_threads=2, device_id=0)
    with pipe:
        enc, _ = fn.readers.file(files=paths)
        pipe.set_outputs(fn.decoders.image_crop(enc, device="mixed", crop=(50, 37), crop_pos_x=0.8, crop_pos_y=0.25))
    (out,) = pipe.run()
    for i, f in enumerate(paths):
        ref = np.asarray(ImageOps.exif_transpose(Image.open(f)).convert("RGB"))
        y0, x0 = O.crop_anchor(0.25, 50, ref.shape[0]), O.crop_anchor(0.8, 37, ref.shape[1])
        assert np.array_equal(out[i].as_cpu(), ref[y0:y0 + 50, x0:x0 + 37]), f"orientation {i + 1}"


def test_image_crop_out_of_bounds_is_an_error(files):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    pipe = Pipeline(batch_size=2, num_threads=2, device_id=0)
    with pipe:
        enc, _ = fn.readers.file(files=files[:2])
        pipe.set_outputs(fn.decoders.image_crop(enc, device="mixed", crop=(1000, 10)))
    with pytest.raises(RuntimeError, match="out of the bounds"):
        pipe.run()


RESIZE_CASES = [
    (dict(size=[100, 150]), dict(size=(100, 150))),
    (dict(resize_x=90), dict(size=(0, 90))),
    (dict(resize_y=77.5), dict(size=(77.5, 0))),
    (dict(resize_x=64, resize_y=0, mode="stretch"), dict(size=(0, 64), mode="stretch")),
    (dict(resize_shorter=80), dict(size=(80, 80), mode="not_smaller")),
    (dict(resize_longer=112), dict(size=(112, 112), mode="not_larger")),
    (dict(resize_shorter=100, max_size=[140]), dict(size=(100, 100), mode="not_smaller", max_size=140)),
    (dict(size=[60, 60], mode="not_larger"), dict(size=(60, 60), mode="not_larger")),
    (dict(size=[50, 70], roi_start=[0.1, 0.2], roi_end=[0.9, 0.7], roi_relative=True),
     dict(size=(50, 70), roi=(0.1, 0.2, 0.9, 0.7), roi_relative=True)),
    (dict(resize_shorter=71.3, subpixel_scale=False), dict(size=(71.3, 71.3), mode="not_smaller", subpixel_scale=False)),
]


@pytest.mark.parametrize("dali_kw,oracle_kw", RESIZE_CASES)
def test_resize_matches_oracle(files, dali_kw, oracle_kw):
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    bs = len(files)
    pipe = Pipeline(batch_size=bs, num_threads=3, device_id=0)
    with pipe:
        enc, _ = fn.readers.file(files=files)
        img = fn.decoders.image(enc, device="mixed")
        pipe.set_outputs(fn.resize(img, interp_type=types.INTERP_TRIANGULAR, **dali_kw))
    (out,) = pipe.run()
    assert "resample" in pipe.executed_kernels()
    for i, r in enumerate(_decoded(files)):
        out_hw, roi = O.resize_params(r.shape[:2], **oracle_kw)
        ref = O.resample_u8(r, out_hw, roi=roi, min_filter=O.FILTER_TRIANGULAR, mag_filter=O.FILTER_TRIANGULAR)
        got = out[i].as_cpu()
        assert got.shape == ref.shape, (i, got.shape, ref.shape)
        assert np.array_equal(got, ref), f"sample {i}: max diff {np.abs(got.astype(int) - ref).max()}"


def test_roi_decode_resize_cmn_pipeline_matches_oracle(files):
    """The validation / NVIDIA-benchmark flavour of the hot path (hw_decoder_bench.py:178-188):
    decoders.image_random_crop -> resize -> crop_mirror_normalize; resize + CMN run as ONE fused kernel."""
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    bs = len(files)
    pipe = Pipeline(batch_size=bs, num_threads=3, device_id=0, prefetch_queue_depth=2)
    with pipe:
        enc, _ = fn.readers.file(files=files)
        img = fn.decoders.image_random_crop(enc, device="mixed", seed=99)
        img = fn.resize(img, size=[64, 80])
        flip = fn.random.coin_flip(probability=0.5, seed=7)
        pipe.set_outputs(fn.crop_mirror_normalize(img, dtype=types.FLOAT16, output_layout="CHW", mean=MEAN, std=STD,
                                                  mirror=flip))
    ref_imgs = _decoded(files)
    mean, inv = O.cmn_norm_args(MEAN, STD)
    for it in range(3):
        (out,) = pipe.run()
        assert "fused_resample_cmn" in pipe.executed_kernels()
        anchors, crops = O.rrc_batch(99, it, [r.shape[:2] for r in ref_imgs])
        mirror = O.coin_flip_batch(7, it, bs, 0.5)
        got = out.as_tensor().cpu().numpy()
        for i in range(bs):
            (y0, x0), (h, w) = anchors[i], crops[i]
            crop = np.ascontiguousarray(ref_imgs[i][y0:y0 + h, x0:x0 + w])
            out_hw, roi = O.resize_params((h, w), (64, 80))
            rs = O.resample_u8(crop, out_hw, roi=roi)
            ref = O.cmn_u8(rs, (0, 0), (64, 80), mirror=bool(mirror[i]), mean=mean, inv_std=inv, layout="CHW", dtype=O.F16)
            assert np.array_equal(got[i].view(np.uint16), ref.view(np.uint16)), (it, i)


def test_resize_flipped_region_matches_oracle(files):
    """roi_start > roi_end flips the axis (resize_attr_base.h:62-96): the region is traversed backwards."""
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    bs = len(files)
    pipe = Pipeline(batch_size=bs, num_threads=3, device_id=0)
    with pipe:
        enc, _ = fn.readers.file(files=files)
        img = fn.decoders.image(enc, device="mixed")
        pipe.set_outputs(fn.resize(img, size=[40, 56], roi_start=[0.9, 0.1], roi_end=[0.2, 0.8], roi_relative=True),
                         fn.resize(img, resize_x=48, roi_start=[0.0, 1.0], roi_end=[1.0, 0.0], roi_relative=True))
    flipped_y, mirrored = pipe.run()
    for i, r in enumerate(_decoded(files)):
        out_hw, roi = O.resize_params(r.shape[:2], size=(40, 56), roi=(0.9, 0.1, 0.2, 0.8), roi_relative=True)
        assert roi[0] > roi[2] and roi[1] < roi[3]
        ref = O.resample_u8(r, out_hw, roi=roi)
        assert np.array_equal(flipped_y[i].as_cpu(), ref), i
        # a whole-image horizontal flip is the mirror image of the plain resize (up to float rounding of the taps)
        out_hw, roi = O.resize_params(r.shape[:2], size=(0, 48), roi=(0.0, 1.0, 1.0, 0.0), roi_relative=True)
        ref = O.resample_u8(r, out_hw, roi=roi)
        got = mirrored[i].as_cpu()
        assert np.array_equal(got, ref), i
        plain = O.resample_u8(r, out_hw, roi=(roi[0], roi[3], roi[2], roi[1]))
        assert np.abs(got.astype(int) - plain[:, ::-1]).max() <= 1


RCM_CASES = [
    dict(resize_kw=dict(resize_shorter=96), crop=(80, 80), mirror=0),
    dict(resize_kw=dict(resize_shorter=96), crop=(80, 64), mirror=1, pos=(0.25, 0.75)),
    dict(resize_kw=dict(size=[90, 120]), crop=(64, 100), mirror=2, pos=(1.0, 0.0)),
    dict(resize_kw=dict(resize_longer=100), crop=(50, 60), mirror=3, rounding="truncate", pos=(0.3, 0.6)),
    dict(resize_kw=dict(resize_x=77), crop=(0, 0), mirror=1),                         # no crop: the whole resized image
    dict(resize_kw=dict(size=[60, 60]), crop=(80, 72), mirror=0),                     # window larger than the image
    # test_resize_crop_mirror.py:75-78: a region that is already flipped in x, stretched, then cropped and mirrored
    dict(resize_kw=dict(size=[100, 130], mode="stretch", roi_start=[0.3, 0.8], roi_end=[0.9, 0.1], roi_relative=True),
         crop=(70, 90), mirror=1, pos=(0.4, 0.2)),
    dict(resize_kw=dict(size=[110, 90], roi_start=[0.7, 0.2], roi_end=[0.1, 0.8], roi_relative=True),
         crop=(100, 64), mirror=2, pos=(0.0, 1.0)),
]


@pytest.mark.parametrize("case", RCM_CASES)
def test_resize_crop_mirror_matches_oracle(files, case):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    bs = len(files)
    kw = dict(case["resize_kw"])
    okw = {}
    if "resize_shorter" in kw:
        okw = dict(size=(kw["resize_shorter"],) * 2, mode="not_smaller")
    elif "resize_longer" in kw:
        okw = dict(size=(kw["resize_longer"],) * 2, mode="not_larger")
    elif "resize_x" in kw:
        okw = dict(size=(0, kw["resize_x"]))
    else:
        okw = dict(size=tuple(kw["size"]))
    if "mode" in kw:
        okw["mode"] = kw["mode"]
    if "roi_start" in kw:
        okw.update(roi=tuple(kw["roi_start"]) + tuple(kw["roi_end"]), roi_relative=True)
    pos = case.get("pos", (0.5, 0.5))
    if case["crop"] != (0, 0):
        kw.update(crop=[float(c) for c in case["crop"]], crop_pos_y=pos[0], crop_pos_x=pos[1])
    pipe = Pipeline(batch_size=bs, num_threads=3, device_id=0)
    with pipe:
        enc, _ = fn.readers.file(files=files)
        img = fn.decoders.image(enc, device="mixed")
        pipe.set_outputs(fn.resize_crop_mirror(img, mirror=case["mirror"], rounding=case.get("rounding", "round"), **kw),
                         fn.resize(img, **case["resize_kw"]))
    out, resized = pipe.run()
    assert out.layout() == "HWC"
    for i, r in enumerate(_decoded(files)):
        out_hw, roi = O.resize_crop_mirror_params(r.shape[:2], crop=case["crop"], crop_pos=pos, mirror=case["mirror"],
                                                  rounding=case.get("rounding", "round"), **okw)
        ref = O.resample_u8(r, out_hw, roi=roi)
        got = out[i].as_cpu()
        assert got.shape == ref.shape, (i, got.shape, ref.shape)
        assert np.array_equal(got, ref), f"sample {i}: max diff {np.abs(got.astype(int) - ref).max()}"
        # the operator's contract: equivalent to resize, then crop, then flip (where the window is inside the image)
        rs = resized[i].as_cpu()
        ch, cw = out_hw
        if ch <= rs.shape[0] and cw <= rs.shape[1]:
            ay = O.crop_anchor(pos[0] if case["crop"][0] > 0 else 0.5, ch, rs.shape[0], case.get("rounding", "round"))
            ax = O.crop_anchor(pos[1] if case["crop"][1] > 0 else 0.5, cw, rs.shape[1], case.get("rounding", "round"))
            seq = rs[ay:ay + ch, ax:ax + cw]
            if case["mirror"] & 1:
                seq = seq[:, ::-1]
            if case["mirror"] & 2:
                seq = seq[::-1]
            assert np.abs(got.astype(int) - seq).max() <= 1, i


def test_resize_crop_mirror_per_sample_mirror_and_fused_normalize(files):
    """Per-sample `mirror` from a DataNode, and ResizeCropMirror -> CropMirrorNormalize runs as one fused kernel."""
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    bs = len(files)
    pipe = Pipeline(batch_size=bs, num_threads=3, device_id=0)
    with pipe:
        enc, _ = fn.readers.file(files=files)
        img = fn.decoders.image(enc, device="mixed")
        flip = fn.random.coin_flip(probability=0.5, seed=5)
        rcm = fn.resize_crop_mirror(img, resize_shorter=72, crop=[64.0, 64.0], mirror=flip)
        pipe.set_outputs(fn.crop_mirror_normalize(rcm, dtype=types.FLOAT, output_layout="CHW", mean=MEAN, std=STD))
    (out,) = pipe.run()
    assert "fused_resample_cmn" in pipe.executed_kernels()
    mirror = O.coin_flip_batch(5, 0, bs, 0.5)
    mean, inv = O.cmn_norm_args(MEAN, STD)
    got = out.as_tensor().cpu().numpy()
    for i, r in enumerate(_decoded(files)):
        out_hw, roi = O.resize_crop_mirror_params(r.shape[:2], crop=(64, 64), mirror=int(mirror[i]), size=(72, 72),
                                                  mode="not_smaller")
        rs = O.resample_u8(r, out_hw, roi=roi)
        ref = O.cmn_u8(rs, (0, 0), (64, 64), mean=mean, inv_std=inv, layout="CHW", dtype=O.F32)
        assert np.array_equal(got[i], ref), i     # the fused kernel rounds to u8 exactly where the two-op chain does


def _llround(v):
    return int(np.floor(v + 0.5)) if v >= 0 else -int(np.floor(-v + 0.5))


@pytest.mark.parametrize("case", ["rel_start_rel_shape", "start_end", "start_shape_axes_hw", "positional_normalized",
                                  "positional_absolute_int"])
def test_image_slice_equals_decode_then_slice(files, case):
    """decoders.image_slice (decoder_schema.cc:200-252, slice_attr.h:40-352): every way of giving the window decodes
    exactly that window; anchor / end are rounded separately (llround) like the reference's CropWindowGenerator."""
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    bs = len(files)
    ref = _decoded(files)
    rng = np.random.default_rng(5)
    pipe = Pipeline(batch_size=bs, num_threads=3, device_id=0, prefetch_queue_depth=1)
    expect = []
    with pipe:
        enc, _ = fn.readers.file(files=files)
        if case == "rel_start_rel_shape":     # default axis order "WH"
            out = fn.decoders.image_slice(enc, device="mixed", rel_start=[0.25, 0.1], rel_shape=[0.5, 0.6])
            for r in ref:
                H, W = r.shape[:2]
                x0, x1 = _llround(0.25 * W), _llround((0.25 + 0.5) * W)
                y0, y1 = _llround(np.float32(0.1) * H), _llround((np.float64(np.float32(0.1)) + np.float64(np.float32(0.6))) * H)
                expect.append(r[y0:y1, x0:x1])
        elif case == "start_end":
            out = fn.decoders.image_slice(enc, device="mixed", start=[10, 5], end=[40, 47])
            expect = [r[5:47, 10:40] for r in ref]
        elif case == "start_shape_axes_hw":
            out = fn.decoders.image_slice(enc, device="mixed", start=[7, 3], shape=[33, 20], axis_names="HW")
            expect = [r[7:40, 3:23] for r in ref]
        elif case == "positional_normalized":
            anchors = rng.uniform(0, 0.4, (bs, 2)).astype(np.float32)
            shapes = rng.uniform(0.2, 0.5, (bs, 2)).astype(np.float32)
            a = fn.external_source(name="a")
            s = fn.external_source(name="s")
            out = fn.decoders.image_slice(enc, a, s, device="mixed")
            for r, an, sh in zip(ref, anchors, shapes):
                H, W = r.shape[:2]
                x0, x1 = _llround(float(an[0]) * W), _llround((float(an[0]) + float(sh[0])) * W)
                y0, y1 = _llround(float(an[1]) * H), _llround((float(an[1]) + float(sh[1])) * H)
                expect.append(r[y0:y1, x0:x1])
        else:
            anchors = rng.integers(0, 20, (bs, 2)).astype(np.int32)
            shapes = rng.integers(8, 28, (bs, 2)).astype(np.int32)
            a = fn.external_source(name="a")
            s = fn.external_source(name="s")
            out = fn.decoders.image_slice(enc, a, s, device="mixed", axis_names="HW")
            expect = [r[an[0]:an[0] + sh[0], an[1]:an[1] + sh[1]] for r, an, sh in zip(ref, anchors, shapes)]
        pipe.set_outputs(out)
    pipe.build()
    if case.startswith("positional"):
        pipe.feed_input("a", anchors)
        pipe.feed_input("s", shapes)
    (res,) = pipe.run()
    for i in range(bs):
        got = res[i].as_cpu()
        assert got.shape == expect[i].shape, (case, i, got.shape, expect[i].shape)
        assert np.array_equal(got, expect[i]), (case, i)


def test_image_slice_argument_errors(files):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    for kw, msg in ((dict(start=[0, 0], rel_start=[0.0, 0.0]), "mutually exclusive"),
                    (dict(end=[5, 5], shape=[5, 5]), "mutually exclusive"),
                    (dict(start=[0, 0], shape=[10000, 4]), "out of the bounds")):
        pipe = Pipeline(batch_size=len(files), num_threads=2, device_id=0, prefetch_queue_depth=1)
        with pipe:
            enc, _ = fn.readers.file(files=files)
            pipe.set_outputs(fn.decoders.image_slice(enc, device="mixed", **kw))
        with pytest.raises(RuntimeError, match=msg):
            pipe.build()
            pipe.run()
