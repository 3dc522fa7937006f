"""Graph-level fusion decoders.image(mixed) -> random_resized_crop (host/pipeline.cpp, TryEnableRoiDecodeFusion): when the
decoder feeds nothing else, only the crop window of every image (plus the reach of the resampling filter) is decoded;
the consumer resamples it with the FULL image's coordinate arithmetic (daliamdResampleArgs.full_h ...).  The pipeline's
output must not change by a bit: against the same pipeline with the fusion switched off (DALI_AMD_ROI_FUSION=0 at build
time of the pipeline), and against the oracle composition for the headline graph.  The reference has the explicit form
only (decoders.image_random_crop + resize, dali/operators/decoder/image/image_decoder_random_crop... /
internal_tools/hw_decoder_bench.py:178-188); this is the same saving for the graph BASELINE.json's metric names."""
import io
import os

import numpy as np
import pytest
from PIL import Image

from oracle import oracle as O
from tests.util import synth_image

pytestmark = pytest.mark.gpu

MEAN = [0.485 * 255, 0.456 * 255, 0.406 * 255]
STD = [0.229 * 255, 0.224 * 255, 0.225 * 255]


def _dataset(root, rng, png=False):
    os.makedirs(os.path.join(root, "0"), exist_ok=True)
    enc = []
    shapes = [(375, 500), (500, 375), (333, 500), (97, 131), (480, 640), (64, 48), (8, 8), (500, 500), (201, 1000), (31, 17)]
    modes = [dict(subsampling="4:2:0"), dict(subsampling="4:4:4"), dict(subsampling="4:2:2"),
             dict(subsampling="4:2:0", restart_marker_rows=2), dict(subsampling="4:2:0", progressive=True)]
    for i, (h, w) in enumerate(shapes):
        buf = io.BytesIO()
        if png and i == 3:
            Image.fromarray(synth_image(rng, h, w)).save(buf, "PNG")
        elif i == 5:
            Image.fromarray(synth_image(rng, h, w)).convert("L").save(buf, "JPEG", quality=80)
        else:
            Image.fromarray(synth_image(rng, h, w)).save(buf, "JPEG", quality=85, **modes[i % len(modes)])
        enc.append(buf.getvalue())
        open(os.path.join(root, "0", f"img_{i:03d}." + ("png" if png and i == 3 else "jpg")), "wb").write(enc[-1])
    return enc


def _run(root, fusion, iters=3, batch=10, cmn=True, extra_consumer=False, **rrc_kw):
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    old = os.environ.get("DALI_AMD_ROI_FUSION")
    os.environ["DALI_AMD_ROI_FUSION"] = "1" if fusion else "0"
    try:
        pipe = Pipeline(batch_size=batch, num_threads=4, device_id=0, seed=77, prefetch_queue_depth=2)
        with pipe:
            jpegs, _ = fn.readers.file(file_root=root, name="Reader")
            images = fn.decoders.image(jpegs, device="mixed", output_type=types.RGB)
            crops = fn.random_resized_crop(images, size=[224, 224], seed=1234, **rrc_kw)
            out = crops
            if cmn:
                out = fn.crop_mirror_normalize(crops, dtype=types.FLOAT16, output_layout="CHW", mean=MEAN, std=STD,
                                               mirror=fn.random.coin_flip(probability=0.5, seed=1235))
            outs = [out] + ([fn.resize(images, size=[32, 32])] if extra_consumer else [])
            pipe.set_outputs(*outs)
        pipe.build()
    finally:
        if old is None:
            os.environ.pop("DALI_AMD_ROI_FUSION", None)
        else:
            os.environ["DALI_AMD_ROI_FUSION"] = old
    res, kernels = [], None
    for _ in range(iters):
        o = pipe.run()
        res.append(o[0].as_tensor().cpu().numpy().copy())
        kernels = pipe.executed_kernels()
    return res, kernels


@pytest.mark.parametrize("rrc_kw", [dict(), dict(interp_type="cubic"), dict(interp_type="lanczos3"), dict(antialias=False),
                                    dict(interp_type="nn"), dict(random_area=[0.9, 1.0]), dict(random_area=[0.01, 0.05]),
                                    dict(size=[64, 300]), dict(size=[600, 600])])
def test_fused_equals_unfused(tmp_path, rrc_kw):
    from dali_amd import types
    kw = dict(rrc_kw)
    if "interp_type" in kw:
        kw["interp_type"] = {"cubic": types.INTERP_CUBIC, "lanczos3": types.INTERP_LANCZOS3, "nn": types.INTERP_NN}[kw["interp_type"]]
    _dataset(str(tmp_path), np.random.default_rng(3), png=True)
    size = kw.pop("size", None)
    # (the size is an argument of _run's graph: patch it through rrc_kw)
    if size:
        kw["size"] = size
    fused, k1 = _run_sized(str(tmp_path), True, **kw)
    plain, k0 = _run_sized(str(tmp_path), False, **kw)
    assert "windows_of_the_consumer" in k1 and "windows_of_the_consumer" not in k0
    for it, (a, b) in enumerate(zip(fused, plain)):
        assert a.dtype == b.dtype and a.shape == b.shape
        assert np.array_equal(a.view(np.uint16), b.view(np.uint16)), (rrc_kw, it, int((a.view(np.uint16) != b.view(np.uint16)).sum()))


def _run_sized(root, fusion, size=None, **kw):
    if size is None:
        return _run(root, fusion, **kw)
    # same graph with another output size
    from dali_amd import fn
    orig = fn.random_resized_crop

    def rrc(images, size=None, **k):
        return orig(images, size=_run_sized.size, **k)
    _run_sized.size = size
    fn.random_resized_crop = rrc
    try:
        return _run(root, fusion, **kw)
    finally:
        fn.random_resized_crop = orig


def test_plain_output_and_oracle(tmp_path):
    """Without the normalisation behind it (u8 HWC output of the resampling kernel itself), against the oracle."""
    enc = _dataset(str(tmp_path), np.random.default_rng(4))
    fused, k1 = _run(str(tmp_path), True, cmn=False, iters=2)
    assert "windows_of_the_consumer" in k1
    shapes = [O.jpeg_decode_rgb(e).shape[:2] for e in enc]
    for it in range(2):
        anchors, crops = O.rrc_batch(1234, it, shapes)
        for i, e in enumerate(enc):
            roi = (anchors[i][0], anchors[i][1], anchors[i][0] + crops[i][0], anchors[i][1] + crops[i][1])
            ref = O.resample_u8(O.jpeg_decode_rgb(e), (224, 224), roi=roi)
            assert np.array_equal(fused[it][i], ref), (it, i)


def test_another_consumer_of_the_images_keeps_the_full_decode(tmp_path):
    _dataset(str(tmp_path), np.random.default_rng(5))
    res, kernels = _run(str(tmp_path), True, extra_consumer=True, iters=1)
    assert "windows_of_the_consumer" not in kernels


def test_exif_orientation(tmp_path):
    """Windows are windows of the UPRIGHT image (adjust_orientation)."""
    rng = np.random.default_rng(6)
    os.makedirs(tmp_path / "0")
    for i, orientation in enumerate([1, 3, 6, 8, 2, 5]):
        buf = io.BytesIO()
        exif = Image.Exif()
        exif[0x0112] = orientation
        Image.fromarray(synth_image(rng, 120 + 16 * i, 200)).save(buf, "JPEG", quality=90, exif=exif)
        (tmp_path / "0" / f"o{i}.jpg").write_bytes(buf.getvalue())
    fused, k1 = _run(str(tmp_path), True, batch=6, iters=2)
    plain, _ = _run(str(tmp_path), False, batch=6, iters=2)
    assert "windows_of_the_consumer" in k1
    for a, b in zip(fused, plain):
        assert np.array_equal(a.view(np.uint16), b.view(np.uint16))


def test_checkpoint_resumes_the_window_sequence(tmp_path):
    """The operator's generator state in a checkpoint is the one BEHIND the windows it has handed out, not behind the ones
    the decoder has already asked for (it runs ahead by the prefetch depth)."""
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    _dataset(str(tmp_path), np.random.default_rng(8))

    def build(checkpoint=None):
        pipe = Pipeline(batch_size=5, num_threads=2, device_id=0, seed=7, prefetch_queue_depth=3, enable_checkpointing=True,
                        checkpoint=checkpoint)
        with pipe:
            jpegs, _ = fn.readers.file(file_root=str(tmp_path), name="Reader")
            pipe.set_outputs(fn.random_resized_crop(fn.decoders.image(jpegs, device="mixed"), size=[96, 96], seed=5))
        pipe.build()
        return pipe
    a = build()
    for _ in range(4):
        a.run()
    assert "windows_of_the_consumer" in a.executed_kernels()
    cp = a.checkpoint()
    want = [a.run()[0].as_tensor().cpu().numpy().copy() for _ in range(3)]
    b = build(checkpoint=cp)
    got = [b.run()[0].as_tensor().cpu().numpy().copy() for _ in range(3)]
    for x, y in zip(want, got):
        assert np.array_equal(x, y)


def test_failed_decode_does_not_shift_the_window_sequence(tmp_path):
    """the windows of an iteration are drawn when the DECODER runs.  A decode that fails behind the draw (here: a
    PNG whose header parses and whose pixel data is cut off) must take the draw back - the crop operator never runs for
    that iteration - so that every later iteration succeeds and gets the windows it gets without the fusion."""
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(11)
    os.makedirs(tmp_path / "0")
    for i in range(12):
        buf = io.BytesIO()
        if i == 5:
            Image.fromarray(synth_image(rng, 90, 120)).save(buf, "PNG")
            data = buf.getvalue()[:200]          # signature + IHDR + the start of IDAT
        else:
            Image.fromarray(synth_image(rng, 100 + 8 * i, 160)).save(buf, "JPEG", quality=85)
            data = buf.getvalue()
        (tmp_path / "0" / f"img_{i:03d}.{'png' if i == 5 else 'jpg'}").write_bytes(data)

    def run(fusion):
        old = os.environ.get("DALI_AMD_ROI_FUSION")
        os.environ["DALI_AMD_ROI_FUSION"] = "1" if fusion else "0"
        try:
            pipe = Pipeline(batch_size=4, num_threads=2, device_id=0, seed=3, prefetch_queue_depth=2)
            with pipe:
                jpegs, _ = fn.readers.file(file_root=str(tmp_path), name="Reader")
                images = fn.decoders.image(jpegs, device="mixed", output_type=types.RGB)
                pipe.set_outputs(fn.random_resized_crop(images, size=[64, 64], seed=99))
            pipe.build()
        finally:
            if old is None:
                os.environ.pop("DALI_AMD_ROI_FUSION", None)
            else:
                os.environ["DALI_AMD_ROI_FUSION"] = old
        res = []
        for _ in range(6):
            try:
                res.append(pipe.run()[0].as_tensor().cpu().numpy().copy())
            except RuntimeError as e:
                assert "img_005" in str(e), str(e)
                res.append(None)
        return res, pipe.executed_kernels()
    fused, k1 = run(True)
    plain, k0 = run(False)
    assert "windows_of_the_consumer" in k1 and "windows_of_the_consumer" not in k0
    assert [r is None for r in fused] == [r is None for r in plain] == [False, True, False, False, True, False]
    for it, (a, b) in enumerate(zip(fused, plain)):
        if a is not None:
            assert np.array_equal(a, b), it
