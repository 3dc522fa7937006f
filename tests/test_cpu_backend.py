"""BASELINE.json configs[0]: the ImageNet train pipe on the CPU backend - decoders.image -> random_resized_crop ->
crop_mirror_normalize, 224 x 224, batch 32 - through dali_amd.Pipeline with NO GPU, checked against the oracle bit
for bit.  The CPU operators are product code (host/jpeg_pixels.cpp, host/host_kernels.cpp: the reference registers
the same three operators for CPU, host_decoder.cc:35-48, random_resized_crop.cc:54, crop_mirror_normalize.cc:83);
nothing under oracle/ is used by them."""
import io
import os

import numpy as np
import pytest
from PIL import Image, ImageOps

from oracle import oracle as O
from tests.util import encode_jpeg, synth_image, synth_jpeg_batch

MEAN = [0.485 * 255, 0.456 * 255, 0.406 * 255]
STD = [0.229 * 255, 0.224 * 255, 0.225 * 255]


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    root = tmp_path_factory.mktemp("cpu_ds")
    rng = np.random.default_rng(1234)
    enc = synth_jpeg_batch(rng, 40, sizes=[(375, 500), (500, 375), (333, 500), (256, 384), (97, 131)])
    files = []
    for i, e in enumerate(enc):
        d = root / f"{i % 4}"
        os.makedirs(d, exist_ok=True)
        (d / f"img_{i:04d}.jpg").write_bytes(e)
    for c in range(4):
        for f in sorted(os.listdir(root / f"{c}")):
            files.append((str(root / f"{c}" / f), c))
    return str(root), files


def test_configs0_cpu_train_pipeline_equals_oracle(dataset):
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    root, files = dataset
    B = 32
    pipe = Pipeline(batch_size=B, num_threads=4, device_id=None, seed=1234, prefetch_queue_depth=2)
    with pipe:
        jpegs, labels = fn.readers.file(file_root=root, name="Reader")
        images = fn.decoders.image(jpegs, device="cpu", output_type=types.RGB)
        crops = fn.random_resized_crop(images, size=[224, 224], seed=1234)
        out = fn.crop_mirror_normalize(crops, dtype=types.FLOAT16, output_layout="CHW", mean=MEAN, std=STD,
                                       mirror=fn.random.coin_flip(probability=0.5, seed=1235))
        pipe.set_outputs(out, labels, images)
    pipe.build()
    mean, inv = O.cmn_norm_args(MEAN, STD)
    for it in range(2):
        data, lab, imgs = pipe.run()
        got = data.as_array()
        picks = [(it * B + i) % len(files) for i in range(B)]
        assert got.shape == (B, 3, 224, 224) and got.dtype == np.float16
        assert list(lab.as_array().reshape(-1)) == [files[k][1] for k in picks]
        enc = [open(files[k][0], "rb").read() for k in picks]
        for i in range(B):
            assert np.array_equal(imgs.at(i), O.jpeg_decode_rgb(enc[i])), (it, i)
        ref = O.pipeline_batch(enc, 1234, 1235, it, mean=mean, inv_std=inv, nthreads=4)
        assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), f"iteration {it}"
        assert {"host_jpeg_decode", "host_resample", "host_cmn"} <= set(pipe.executed_kernels())


def test_cpu_decoder_orientation_and_raster_formats(tmp_path):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(2)
    files = []
    for o in range(1, 9):
        img = Image.fromarray(synth_image(rng, 40 + o, 64))
        exif = Image.Exif()
        exif[0x0112] = o
        p = tmp_path / f"o{o}.jpg"
        img.save(p, "JPEG", quality=90, exif=exif)
        files.append(str(p))
    p = tmp_path / "x.png"
    Image.fromarray(synth_image(rng, 33, 21)).save(p, "PNG")
    files.append(str(p))
    p = tmp_path / "g.jpg"
    p.write_bytes(encode_jpeg(synth_image(rng, 50, 70, 1), 80, progressive=True))
    files.append(str(p))
    for adjust in (True, False):
        pipe = Pipeline(batch_size=len(files), num_threads=2, device_id=None)
        with pipe:
            enc, _ = fn.readers.file(files=files)
            pipe.set_outputs(fn.decoders.image(enc, device="cpu", adjust_orientation=adjust))
        (out,) = pipe.run()
        assert out.layout() == "HWC"
        for i, f in enumerate(files):
            im = Image.open(f)
            ref = np.asarray((ImageOps.exif_transpose(im) if adjust else im).convert("RGB"))
            assert np.array_equal(out.at(i), ref), (i, adjust)


def test_cpu_resize_and_stand_alone_cmn_match_oracle(dataset):
    """fn.resize and a cropping crop_mirror_normalize on the CPU backend (device inferred from the CPU input)."""
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    root, files = dataset
    B = 8
    pipe = Pipeline(batch_size=B, num_threads=3, device_id=None, prefetch_queue_depth=1)
    with pipe:
        jpegs, _ = fn.readers.file(file_root=root)
        images = fn.decoders.image(jpegs, device="cpu")
        small = fn.resize(images, resize_shorter=160)
        out = fn.crop_mirror_normalize(small, dtype=types.FLOAT, output_layout="HWC", crop=(128, 112), mean=MEAN, std=STD,
                                       crop_pos_x=0.25, crop_pos_y=0.75, mirror=1)
        pipe.set_outputs(out, small)
    (got, small_out) = pipe.run()
    mean, inv = O.cmn_norm_args(MEAN, STD)
    for i in range(B):
        img = O.jpeg_decode_rgb(open(files[i][0], "rb").read())
        out_hw, roi = O.resize_params(img.shape[:2], size=(160, 160), mode="not_smaller")   # = resize_shorter=160
        r = small_out.at(i)
        assert r.shape[:2] == tuple(out_hw), (i, r.shape, out_hw)
        assert np.array_equal(r, O.resample_u8(img, out_hw, roi=roi)), i
        ay, ax = int(O.crop_anchor(0.75, 128, r.shape[0])), int(O.crop_anchor(0.25, 112, r.shape[1]))
        ref = O.cmn_u8(r, (ay, ax), (128, 112), mirror=True, mean=mean, inv_std=inv, dtype=O.F32, layout="HWC")
        assert np.array_equal(got.at(i), ref), i


def test_cpu_resize_every_filter_matches_oracle(dataset):
    """fn.resize on the CPU backend with nearest neighbour and the tabulated windows: product host code == oracle."""
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    root, files = dataset
    B = 4
    for name, of in [("INTERP_NN", O.FILTER_NN), ("INTERP_CUBIC", O.FILTER_CUBIC), ("INTERP_LANCZOS3", O.FILTER_LANCZOS3),
                     ("INTERP_GAUSSIAN", O.FILTER_GAUSSIAN)]:
        for size in ((100, 150), (400, 600)):
            pipe = Pipeline(batch_size=B, num_threads=2, device_id=None, prefetch_queue_depth=1)
            with pipe:
                jpegs, _ = fn.readers.file(file_root=root)
                images = fn.decoders.image(jpegs, device="cpu")
                pipe.set_outputs(fn.resize(images, size=list(size), interp_type=getattr(types, name)))
            (got,) = pipe.run()
            for i in range(B):
                img = O.jpeg_decode_rgb(open(files[i][0], "rb").read())
                ref = O.resample_u8(img, size, min_filter=of, mag_filter=of)
                assert np.array_equal(got.at(i), ref), (name, size, i)


def test_cpu_resize_other_element_types_match_oracle():
    """fn.resize on i16 / u16 / f32 images and with dtype=FLOAT (the unrounded result): the typed two-pass path of the
    host backend against the oracle, bit for bit (SIMD-store lanes = 16 bytes / sizeof(Out) decide the rounding regions)."""
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(3)
    base = [rng.normal(0, 1, (h, w, c)) for h, w, c in [(60, 90, 3), (33, 47, 1), (120, 80, 3)]]
    cases = [(np.int16, lambda a: np.clip(a * 9000, -32768, 32767)), (np.uint16, lambda a: np.clip(a * 9000 + 30000, 0, 65535)),
             (np.float32, lambda a: a * 3), (np.uint8, lambda a: np.clip(a * 60 + 128, 0, 255))]
    for np_t, conv in cases:
        imgs = [conv(b).astype(np_t) for b in base]
        for size, dtype in [((40, 50), None), ((150, 70), None), ((40, 50), types.FLOAT)]:
            if np_t == np.uint8 and dtype is None:
                continue       # the plain u8 path has its own tests
            pipe = Pipeline(batch_size=len(imgs), num_threads=2, device_id=None, prefetch_queue_depth=1)
            with pipe:
                x = fn.external_source(name="x", layout="HWC")
                kw = {} if dtype is None else {"dtype": dtype}
                pipe.set_outputs(fn.resize(x, size=list(size), **kw))
            pipe.feed_input("x", imgs, layout="HWC")
            (out,) = pipe.run()
            for i, im in enumerate(imgs):
                ref = O.resample_typed(im, size, out_type=O.T_F32 if dtype is not None else None)
                got = out.at(i)
                assert got.dtype == ref.dtype and got.shape == ref.shape, (np_t, size, got.dtype, ref.dtype)
                assert np.array_equal(got, ref), (np_t, size, dtype, i, np.abs(got.astype(np.float64) - ref).max())


def test_scalar_and_avx2_idct_give_the_oracle_bytes(tmp_path):
    """decoders.image(device="cpu") picks the eight-lanes-per-register inverse DCT where the CPU has AVX2; the scalar
    loop it replaces must stay the same function (DALI_AMD_HOST_NO_AVX2=1 selects it: a fresh process, the choice is
    made once).  Both against the oracle on 4:2:0 / 4:4:4 / 4:2:2 / gray streams of ragged sizes."""
    import subprocess
    import sys
    rng = np.random.default_rng(77)
    cases = [((37, 53), "4:2:0", 75), ((64, 48), "4:4:4", 90), ((121, 200), "4:2:2", 60), ((50, 70), "4:2:0", 98)]
    paths = []
    for i, ((h, w), sub, q) in enumerate(cases):
        p = tmp_path / f"s{i}.jpg"
        p.write_bytes(encode_jpeg(synth_image(rng, h, w), q, subsampling=sub))
        paths.append(str(p))
    g = tmp_path / "g.jpg"
    buf = io.BytesIO()
    Image.fromarray(synth_image(rng, 45, 67)[:, :, 0]).save(buf, "JPEG", quality=80)
    g.write_bytes(buf.getvalue())
    paths.append(str(g))
    script = (
        "import sys, numpy as np\n"
        "from dali_amd import fn, types\n"
        "from dali_amd.pipeline import Pipeline\n"
        "files = sys.argv[2:]\n"
        "pipe = Pipeline(batch_size=len(files), num_threads=2, device_id=None)\n"
        "with pipe:\n"
        "    enc, _ = fn.readers.file(files=files)\n"
        "    pipe.set_outputs(fn.decoders.image(enc, device='cpu', output_type=types.RGB))\n"
        "pipe.build()\n"
        "(out,) = pipe.run()\n"
        "np.savez(sys.argv[1], *[np.asarray(out.at(i)) for i in range(len(files))])\n")
    for tag, env in (("avx2", {}), ("scalar", {"DALI_AMD_HOST_NO_AVX2": "1"})):
        dst = str(tmp_path / f"{tag}.npz")
        e = dict(os.environ, **env)
        e["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + e.get("PYTHONPATH", "")
        subprocess.run([sys.executable, "-c", script, dst] + paths, check=True, env=e, timeout=300)
        got = np.load(dst)
        for i, p in enumerate(paths):
            ref = O.jpeg_decode_rgb(open(p, "rb").read())
            assert np.array_equal(got[f"arr_{i}"], ref), (tag, p)
