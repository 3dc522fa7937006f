"""GPU: decoders.image on batches that mix JPEG with the loss-less container formats (PNG, BMP, PNM).  The JPEGs take
the GPU decoder, the others the host decoders (tests/test_image_formats.py pins those against Pillow) + one upload."""
import io

import numpy as np
import pytest
from PIL import Image

from oracle import oracle as O
from tests.util import encode_jpeg, synth_image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mixed(tmp_path_factory):
    root = tmp_path_factory.mktemp("formats")
    rng = np.random.default_rng(17)
    files, refs = [], []

    def add(name, data, ref):
        p = root / name
        p.write_bytes(data)
        files.append(str(p))
        refs.append(ref)

    def save(img, fmt, **kw):
        b = io.BytesIO()
        img.save(b, fmt, **kw)
        return b.getvalue()

    for k, (h, w) in enumerate([(120, 160), (75, 211), (64, 48)]):
        enc = encode_jpeg(synth_image(rng, h, w), 85, subsampling="4:2:0")
        add(f"a{k}.jpg", enc, O.jpeg_decode_rgb(enc))
    a = synth_image(rng, 90, 130)
    add("b.png", save(Image.fromarray(a), "PNG"), a)
    g = synth_image(rng, 50, 70, 1).reshape(50, 70)
    add("c_gray.png", save(Image.fromarray(g), "PNG"), np.repeat(g[..., None], 3, 2))
    pal = Image.fromarray(a).quantize(64)
    add("d_palette.png", save(pal, "PNG"), np.asarray(pal.convert("RGB")))
    b = synth_image(rng, 61, 77)
    add("e.bmp", save(Image.fromarray(b), "BMP"), b)
    add("f.ppm", save(Image.fromarray(b[:40, :33]), "PPM"), b[:40, :33])
    return files, refs


def test_mixed_format_batches(mixed):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    files, refs = mixed
    bs = len(files)
    pipe = Pipeline(batch_size=bs, num_threads=3, device_id=0, prefetch_queue_depth=2)
    with pipe:
        enc, _ = fn.readers.file(files=files)
        img = fn.decoders.image(enc, device="mixed")
        pipe.set_outputs(img, fn.resize(img, size=[32, 40]))
    for it in range(3):
        out, small = pipe.run()
        assert "jpeg_huffman" in pipe.executed_kernels()
        for i in range(bs):
            got = out[i].as_cpu()
            assert got.shape == refs[i].shape, (i, got.shape)
            assert np.array_equal(got, refs[i]), (it, files[i])
            ref_small = O.resample_u8(np.ascontiguousarray(refs[i]), (32, 40))
            assert np.array_equal(small[i].as_cpu(), ref_small), (it, files[i])


def test_batch_without_any_jpeg_and_fixed_crop(mixed):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    files, refs = mixed
    sel = [3, 4, 6, 7]                                    # png, gray png, bmp, ppm
    pipe = Pipeline(batch_size=len(sel), num_threads=2, device_id=0)
    with pipe:
        enc, _ = fn.readers.file(files=[files[i] for i in sel])
        pipe.set_outputs(fn.decoders.image(enc, device="mixed"),
                         fn.decoders.image_crop(enc, device="mixed", crop=(30, 25), crop_pos_x=0.25, crop_pos_y=1.0))
    full, crop = pipe.run()
    assert pipe.executed_kernels() == []                   # nothing for the GPU decoder to do
    for j, i in enumerate(sel):
        assert np.array_equal(full[j].as_cpu(), refs[i])
        H, W = refs[i].shape[:2]
        y0, x0 = O.crop_anchor(1.0, 30, H), O.crop_anchor(0.25, 25, W)
        assert np.array_equal(crop[j].as_cpu(), refs[i][y0:y0 + 30, x0:x0 + 25]), files[i]


def test_broken_png_is_reported_with_its_file_name(mixed, tmp_path):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    files, _ = mixed
    bad = tmp_path / "broken.png"
    data = bytearray(open(files[3], "rb").read())
    data[len(data) // 2] ^= 0xFF
    bad.write_bytes(bytes(data))
    pipe = Pipeline(batch_size=2, num_threads=2, device_id=0)
    with pipe:
        enc, _ = fn.readers.file(files=[files[0], str(bad)])
        pipe.set_outputs(fn.decoders.image(enc, device="mixed"))
    with pytest.raises(RuntimeError, match=r"Failed to decode .*broken\.png: PNG: (CRC error|corrupt)"):
        pipe.run()
    gif = tmp_path / "x.gif"
    gif.write_bytes(b"GIF89a" + bytes(64))
    pipe = Pipeline(batch_size=1, num_threads=1, device_id=0)
    with pipe:
        enc, _ = fn.readers.file(files=[str(gif)])
        pipe.set_outputs(fn.decoders.image(enc, device="mixed"))
    with pytest.raises(RuntimeError, match="unrecognised image format"):
        pipe.run()
