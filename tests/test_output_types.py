"""decoders.image `output_type` (RGB / BGR / GRAY / YCbCr / ANY_DATA) and four-component (CMYK / YCCK) JPEG streams,
on the CPU backend here and on the mixed backend under -m gpu.

Pins: GRAY of a YCbCr stream = libjpeg-turbo's own grayscale output (Pillow `draft("L")`), which is what the reference
asks nvImageCodec for (image_decoder.h:538-541); BGR / YCbCr = ConvertCPU's formulas on the RGB result
(operators/imgcodec/util/convert.h:140-192, color_space_conversion_impl.h:62-103); CMYK / YCCK -> RGB = Pillow
(the reference's conversion lives in un-vendored nvImageCodec)."""
import io

import numpy as np
import pytest
from PIL import Image

from tests.util import encode_jpeg, synth_image


def _ycbcr601(rgb):
    f = rgb.astype(np.float32)
    r, g, b = f[..., 0], f[..., 1], f[..., 2]
    c = np.float32
    y = c(0.25678823529) * r + c(0.50412941176) * g + c(0.09790588235) * b + c(16)
    cb = c(-0.14822289945) * r + c(-0.29099278682) * g + c(0.43921568627) * b + c(128)
    cr = c(0.43921568627) * r + c(-0.36778831435) * g + c(-0.07142737192) * b + c(128)
    return np.clip(np.floor(np.stack([y, cb, cr], -1) + c(0.5)), 0, 255).astype(np.uint8)


def _gray(rgb):
    f = rgb.astype(np.float32)
    c = np.float32
    return np.clip(np.floor(c(0.299) * f[..., 0] + c(0.587) * f[..., 1] + c(0.114) * f[..., 2] + c(0.5)), 0, 255).astype(np.uint8)


@pytest.fixture(scope="module")
def streams(tmp_path_factory):
    root = tmp_path_factory.mktemp("otypes")
    rng = np.random.default_rng(12)
    items = []   # (path, kind)
    for k, (hw, sub) in enumerate([((75, 100), "4:2:0"), ((64, 48), "4:4:4"), ((97, 131), "4:2:2")]):
        p = root / f"c{k}.jpg"
        p.write_bytes(encode_jpeg(synth_image(rng, *hw), 85, sub))
        items.append((str(p), "ycc"))
    p = root / "gray.jpg"
    p.write_bytes(encode_jpeg(synth_image(rng, 50, 70, 1), 80))
    items.append((str(p), "gray"))
    cm = Image.fromarray(synth_image(rng, 60, 90)).convert("CMYK")
    b = io.BytesIO()
    cm.save(b, "JPEG", quality=90)
    p = root / "cmyk.jpg"
    p.write_bytes(b.getvalue())
    items.append((str(p), "cmyk"))
    d = bytearray(b.getvalue())
    d[d.index(b"Adobe") + 11] = 2          # the same samples declared YCCK (Adobe transform 2)
    p = root / "ycck.jpg"
    p.write_bytes(bytes(d))
    items.append((str(p), "cmyk"))
    p = root / "x.png"
    Image.fromarray(synth_image(rng, 33, 21)).save(p, "PNG")
    items.append((str(p), "raster"))
    return items


def _expected(path, kind, out_type):
    from dali_amd import types
    im = Image.open(path)
    rgb = np.asarray(im.convert("RGB"))
    if out_type == types.RGB or out_type == types.ANY_DATA and kind != "gray":
        return rgb
    if out_type == types.BGR:
        return rgb[:, :, ::-1]
    if out_type == types.YCbCr:
        return _ycbcr601(rgb)
    # GRAY (or ANY_DATA of a grayscale stream): libjpeg's luma plane for gray / YCbCr streams, the formula otherwise
    if kind in ("ycc", "gray"):
        im2 = Image.open(path)
        im2.draft("L", im2.size)
        return np.asarray(im2.convert("L"))[:, :, None]
    return _gray(rgb)[:, :, None]


def _run(items, device, out_type):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    files = [p for p, _ in items]
    pipe = Pipeline(batch_size=len(files), num_threads=3, device_id=0 if device == "mixed" else None, prefetch_queue_depth=1)
    with pipe:
        enc, _ = fn.readers.file(files=files)
        pipe.set_outputs(fn.decoders.image(enc, device=device, output_type=out_type))
    (out,) = pipe.run()
    return [out[i].as_cpu() if device == "mixed" else out.at(i) for i in range(len(files))]


@pytest.mark.parametrize("name", ["RGB", "BGR", "GRAY", "YCbCr", "ANY_DATA"])
def test_cpu_decoder_output_types_and_cmyk(streams, name):
    from dali_amd import types
    out_type = getattr(types, name)
    got = _run(streams, "cpu", out_type)
    for (path, kind), g in zip(streams, got):
        ref = _expected(path, kind, out_type)
        assert g.shape == ref.shape, (path, name, g.shape, ref.shape)
        assert np.array_equal(g, ref), (path, name, int(np.abs(g.astype(int) - ref.astype(int)).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["RGB", "BGR", "GRAY", "YCbCr"])
def test_mixed_decoder_output_types_and_cmyk(streams, name):
    """The same streams through decoders.image(device="mixed"): GPU entropy decode + colour kernel with the output
    format, four-component and PNG samples decoded on the host and uploaded next to them."""
    from dali_amd import types
    out_type = getattr(types, name)
    got = _run(streams, "mixed", out_type)
    for (path, kind), g in zip(streams, got):
        ref = _expected(path, kind, out_type)
        assert g.shape == ref.shape, (path, name, g.shape, ref.shape)
        assert np.array_equal(g, ref), (path, name, int(np.abs(g.astype(int) - ref.astype(int)).max()))
