"""GPU: the encoded-stream cache of the image decoders (`cache_type="encoded"`, an MI355X extension of the reference's
decoder cache) together with the readers' `skip_cached_images` (loader.h:466-480): from the second epoch on the reader
emits EMPTY samples and the decoder decodes from the segments resident in HBM - the pixels must be those of a fresh
decode, for whole images, region-of-interest decodes and batches that mix resident, new, progressive and PNG samples.
Round 6: a progressive stream (host entropy decoder) is kept too - as the lossless baseline re-encoding of its coefficients
(daliamdJpegEncodeBaselineScan), which the device decodes from then on; DALI_AMD_TRANSCODE_PROGRESSIVE=0 switches that off."""
import gc

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import encode_jpeg, synth_image

pytestmark = pytest.mark.gpu

SIZES = [(120, 160), (200, 150), (97, 131), (240, 320), (64, 48), (333, 500), (180, 180), (75, 211)]


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    root = tmp_path_factory.mktemp("ecache")
    rng = np.random.default_rng(15)
    out = []
    for i, hw in enumerate(SIZES):
        kw = dict(subsampling=["4:2:0", "4:4:4", "4:2:2"][i % 3])
        if i == 3:
            kw["progressive"] = True          # host entropy decoder; kept resident re-encoded as a baseline stream
        p = root / f"img{i}.jpg"
        p.write_bytes(encode_jpeg(synth_image(rng, *hw), 85, **kw))
        out.append(str(p))
    return out


@pytest.fixture(scope="module")
def decoded(files):
    return [O.jpeg_decode_rgb(open(f, "rb").read()) for f in files]


CACHE_TYPE = ["encoded"]


@pytest.fixture(autouse=True, params=["encoded", "indexed"])
def _collect(request):
    # every test with both forms of residency: the segments as they are in the file, and (round 5) the un-stuffed stream
    # with the decoder state in front of every slice, from which later epochs decode without parsing it again
    CACHE_TYPE[0] = request.param
    gc.collect()
    yield
    gc.collect()


def _pipe(files, batch, decoder="image", skip=True, outputs="image", **decoder_kw):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    pipe = Pipeline(batch_size=batch, num_threads=3, device_id=0, prefetch_queue_depth=2, seed=11)
    with pipe:
        enc, _ = fn.readers.file(files=files, skip_cached_images=skip)
        img = getattr(fn.decoders, decoder)(enc, device="mixed", cache_size=64, cache_type=CACHE_TYPE[0], **decoder_kw)
        pipe.set_outputs(*((img, enc) if outputs == "both" else (img,)))
    return pipe


def test_second_epoch_decodes_from_hbm_without_reading_the_files(files, decoded):
    pipe = _pipe(files, 4, outputs="both")
    for it in range(8):                                   # 4 epochs of 2 iterations
        img, enc = pipe.run()
        for i in range(4):
            k = (4 * it + i) % len(files)
            assert np.array_equal(img[i].as_cpu(), decoded[k]), (it, i)
            nbytes = enc.at(i).size
            # (epoch 1 fills; the reader runs up to two batches ahead of the decoder; a stream kept with its index becomes
            # resident behind the launch of the decode that builds it, a stream kept as it is in front of it)
            if it >= (3 if CACHE_TYPE[0] == "encoded" else 4):
                assert nbytes == 0, f"iteration {it}: sample {k} was read again ({nbytes} bytes)"
        assert "jpeg_huffman" in pipe.executed_kernels()   # a hit is decoded anew
        if it >= 4:
            assert ("jpeg_huffman_indexed" in pipe.executed_kernels()) == (CACHE_TYPE[0] == "indexed")


@pytest.mark.parametrize("zero_copy", ["0", "1"])
def test_files_may_disappear_once_resident(tmp_path, decoded, files, monkeypatch, zero_copy):
    # Both readers: the copying one and the one that hands out page-locked resident copies for a device-side fetch
    # (DALI_AMD_READER_ZERO_COPY=1, the default for a process with at most four CPUs).  Round 5's zero-copy reader registered
    # the file MAPPINGS with the device and this very test stalled for minutes when the files were truncated (the driver
    # evicted the process's queues); round 6 keeps anonymous page-locked copies, which no change to the file can touch.
    monkeypatch.setenv("DALI_AMD_READER_ZERO_COPY", zero_copy)
    import shutil
    mine = []
    for i in (0, 1, 2, 4):
        dst = tmp_path / f"c{i}.jpg"
        shutil.copy(files[i], dst)
        mine.append(str(dst))
    pipe = _pipe(mine, 4)
    for it in range(3):                                   # (the reader is a few batches ahead: let it catch up with the cache)
        pipe.run()
    sizes = [open(f, "rb").read() for f in mine]
    for f in mine:
        open(f, "wb").close()                             # truncate: a read of it now yields an invalid stream
    for it in range(4):
        (img,) = pipe.run()
        for j, i in enumerate((0, 1, 2, 4)):
            assert np.array_equal(img[j].as_cpu(), decoded[i]), (it, i)
    assert all(len(s) > 0 for s in sizes)


def test_roi_decoders_decode_windows_from_the_resident_streams(files, decoded):
    base = [f for k, f in enumerate(files) if k != 3]
    ref = [d for k, d in enumerate(decoded) if k != 3]
    pipe = _pipe(base, len(base), decoder="image_random_crop", random_area=[0.2, 0.8], seed=1234, outputs="both")
    for it in range(5):
        out, enc = pipe.run()
        anchors, crops = O.rrc_batch(1234, it, [r.shape[:2] for r in ref], area=(0.2, 0.8))
        for i, r in enumerate(ref):
            (y0, x0), (h, w) = anchors[i], crops[i]
            assert np.array_equal(out[i].as_cpu(), r[y0:y0 + h, x0:x0 + w]), (it, i)
    assert all(enc.at(i).size == 0 for i in range(len(base)))


def test_mixed_batches_resident_new_and_other_formats(tmp_path, files, decoded):
    """Epoch 2 of a data set that grew: resident samples, a stream the cache has not seen, the progressive one and a
    PNG in one batch."""
    import io
    from PIL import Image
    png = tmp_path / "extra.png"
    rng = np.random.default_rng(3)
    pix = synth_image(rng, 50, 70)
    Image.fromarray(pix).save(png)
    first = [files[0], files[1], files[3], files[5]]
    pipe = _pipe(first, 4)
    for it in range(3):
        (img,) = pipe.run()
    del pipe
    # (the cache lives as long as a pipeline holds it: build the second pipeline before dropping the first one)
    pipe1 = _pipe(first, 4)
    pipe1.run()
    mixed = [files[0], files[2], str(png), files[3], files[5], files[6]]
    pipe2 = _pipe(mixed, 6, outputs="both")
    want = [decoded[0], decoded[2], pix, decoded[3], decoded[5], decoded[6]]
    for it in range(6):
        img, enc = pipe2.run()
        for i in range(6):
            assert np.array_equal(img[i].as_cpu(), want[i]), (it, i)
        if it == 0:
            assert enc.at(2).size > 0                          # the PNG's first sighting: read, decoded on the host
    assert enc.at(0).size == 0 and enc.at(4).size == 0        # resident from pipe1's epoch
    assert enc.at(3).size == 0                                 # the progressive JPEG: resident re-encoded since pipe1's epoch
    assert enc.at(2).size == 0                                 # round 6: the PNG is resident as its decoded image ("raster resident")


def test_raster_residents_serve_windows_as_views(tmp_path):
    """CMYK JPEG + PNG + BMP through the region-of-interest decoder with the encoded cache: first sighting decodes the whole
    image into the cache slot, every later epoch hands out a window of the resident image - no file read, no host decode."""
    import io
    from PIL import Image
    rng = np.random.default_rng(21)
    files, ref = [], []
    pix = synth_image(rng, 90, 120)
    cm = Image.fromarray(pix).convert("CMYK")
    b = io.BytesIO()
    cm.save(b, "JPEG", quality=90)
    (tmp_path / "a_cmyk.jpg").write_bytes(b.getvalue())
    files.append(str(tmp_path / "a_cmyk.jpg"))
    ref.append(np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGB")))
    for name in ("b.png", "c.bmp"):
        pix = synth_image(rng, 70, 101)
        Image.fromarray(pix).save(tmp_path / name)
        files.append(str(tmp_path / name))
        ref.append(pix)
    jp = encode_jpeg(synth_image(rng, 64, 80), 85)
    (tmp_path / "d.jpg").write_bytes(jp)
    files.append(str(tmp_path / "d.jpg"))
    ref.append(O.jpeg_decode_rgb(jp))
    pipe = _pipe(files, 4, decoder="image_random_crop", random_area=[0.2, 0.8], seed=1234, outputs="both")
    for it in range(6):
        out, enc = pipe.run()
        anchors, crops = O.rrc_batch(1234, it, [r.shape[:2] for r in ref], area=(0.2, 0.8))
        for i, r in enumerate(ref):
            (y0, x0), (h, w) = anchors[i], crops[i]
            assert np.array_equal(out[i].as_cpu(), r[y0:y0 + h, x0:x0 + w]), (it, i)
    assert all(enc.at(i).size == 0 for i in range(4)), [enc.at(i).size for i in range(4)]


def test_progressive_streams_stay_host_decoded_when_the_reencoding_is_switched_off(files, decoded, monkeypatch):
    monkeypatch.setenv("DALI_AMD_TRANSCODE_PROGRESSIVE", "0")
    pipe = _pipe([files[3], files[0]], 2, outputs="both")
    for it in range(6):
        img, enc = pipe.run()
        assert np.array_equal(img[0].as_cpu(), decoded[3]) and np.array_equal(img[1].as_cpu(), decoded[0]), it
        assert enc.at(0).size > 0, "not kept: read and decoded on the host every epoch"
    assert enc.at(1).size == 0
    assert "jpeg_idct" in pipe.executed_kernels()              # the stand-alone IDCT of host-decoded coefficients still runs


def test_without_skip_the_reader_still_reads_but_the_decoder_uses_the_cache(files, decoded):
    pipe = _pipe(files, 8, skip=False, outputs="both")
    for it in range(3):
        img, enc = pipe.run()
        for i in range(8):
            assert np.array_equal(img[i].as_cpu(), decoded[i]), (it, i)
            assert enc.at(i).size > 0


def test_cache_too_small_keeps_what_fits_and_decodes_the_rest(files, decoded):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(9)
    import tempfile, os
    d = tempfile.mkdtemp()
    big, ref = [], []
    for i in range(6):
        img = rng.integers(0, 256, (300, 400, 3), dtype=np.uint8)    # noise: ~290 KB per stream at q95: three fit in 1 MB
        enc = encode_jpeg(img, 95, subsampling="4:4:4")
        p = os.path.join(d, f"n{i}.jpg")
        open(p, "wb").write(enc)
        big.append(p)
        ref.append(O.jpeg_decode_rgb(enc))
    pipe = Pipeline(batch_size=6, num_threads=3, device_id=0, prefetch_queue_depth=1, seed=1)
    with pipe:
        enc, _ = fn.readers.file(files=big, skip_cached_images=True)
        pipe.set_outputs(fn.decoders.image(enc, device="mixed", cache_size=1, cache_type=CACHE_TYPE[0]), enc)
    for it in range(4):
        img, e = pipe.run()
        for i in range(6):
            assert np.array_equal(img[i].as_cpu(), ref[i]), (it, i)
    sizes = [e.at(i).size for i in range(6)]
    assert 0 < sum(s == 0 for s in sizes) < 6, sizes          # some resident, some read every epoch


def test_flat_streams_are_kept_with_their_index_whatever_the_cache_type(tmp_path):
    """Content that compresses to a few bits per block (large flat areas) never re-synchronises inside a 256-byte slice: the
    position pass of the entropy decoder then relaxes lane by lane.  Such a stream becomes resident WITH its index under
    cache_type="encoded" too and is decoded from it from the second epoch on (an ordinary stream next to it is not)."""
    if CACHE_TYPE[0] != "encoded":
        pytest.skip("the other cache type indexes everything")
    rng = np.random.default_rng(17)
    flat = np.full((480, 640, 3), (250, 250, 250), np.uint8)
    flat[200:260, 300:380] = synth_image(rng, 60, 80)                  # a small textured patch in a blank frame
    a = encode_jpeg(flat, 85, subsampling="4:2:0")
    b = encode_jpeg(synth_image(rng, 240, 320), 85, subsampling="4:2:0")
    assert len(a) * 8 < 64 * (30 * 40 * 6)                             # fewer than 64 bits per block
    for name, data in (("flat.jpg", a), ("busy.jpg", b)):
        (tmp_path / name).write_bytes(data)
    ref = {"flat.jpg": O.jpeg_decode_rgb(a), "busy.jpg": O.jpeg_decode_rgb(b)}
    for name, indexed in (("flat.jpg", True), ("busy.jpg", False)):
        pipe = _pipe([str(tmp_path / name)], 1, outputs="both")
        for it in range(6):
            img, enc = pipe.run()
            assert np.array_equal(img[0].as_cpu(), ref[name]), (name, it)
        assert enc.at(0).size == 0
        assert ("jpeg_huffman_indexed" in pipe.executed_kernels()) == indexed, name
        del pipe
        gc.collect()


def test_a_full_cache_leaves_progressive_and_raster_samples_as_host_decodes(tmp_path):
    """The re-encoded progressive stream and the decoded PNG / CMYK images need room like any other resident; when the blob is
    full they stay what they were - host decodes every epoch - and the batches do not change."""
    import io
    from PIL import Image
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(19)
    files, ref = [], []
    noise = rng.integers(0, 256, (420, 560, 3), dtype=np.uint8)
    a = encode_jpeg(noise, 97, subsampling="4:4:4")                      # ~0.6 MB: takes most of the 1 MB cache
    (tmp_path / "0_big.jpg").write_bytes(a)
    files.append(str(tmp_path / "0_big.jpg")); ref.append(O.jpeg_decode_rgb(a))
    b = encode_jpeg(rng.integers(0, 256, (300, 400, 3), dtype=np.uint8), 95, subsampling="4:4:4", progressive=True)
    (tmp_path / "1_prog.jpg").write_bytes(b)
    files.append(str(tmp_path / "1_prog.jpg")); ref.append(O.jpeg_decode_rgb(b))
    pix = rng.integers(0, 256, (400, 500, 3), dtype=np.uint8)            # 0.6 MB decoded
    Image.fromarray(pix).save(tmp_path / "2_img.png")
    files.append(str(tmp_path / "2_img.png")); ref.append(pix)
    cm = Image.fromarray(synth_image(rng, 300, 420)).convert("CMYK")
    buf = io.BytesIO(); cm.save(buf, "JPEG", quality=90)
    (tmp_path / "3_cmyk.jpg").write_bytes(buf.getvalue())
    files.append(str(tmp_path / "3_cmyk.jpg")); ref.append(np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB")))
    pipe = Pipeline(batch_size=4, num_threads=3, device_id=0, prefetch_queue_depth=2, seed=1)
    with pipe:
        enc, _ = fn.readers.file(files=files, skip_cached_images=True)
        pipe.set_outputs(fn.decoders.image(enc, device="mixed", cache_size=1, cache_type=CACHE_TYPE[0]), enc)
    for it in range(6):
        img, e = pipe.run()
        for i in range(4):
            assert np.array_equal(img[i].as_cpu(), ref[i]), (it, i)
    sizes = [e.at(i).size for i in range(4)]
    # (what fits is resident - the samples that reserve first: the decoded rasters -, the rest found no room and is read and
    # decoded every epoch)
    assert 0 < sum(s == 0 for s in sizes) < 4, sizes


def test_windows_from_reencoded_and_raster_residents(tmp_path, files, decoded):
    """Region-of-interest decoding over residents of every kind at once: a baseline stream, the progressive one (resident as its
    baseline re-encoding: the window is decoded on the device from the second epoch on), a PNG (resident decoded: the window is a
    view).  Every window of every epoch equals the crop of the oracle's image."""
    from PIL import Image
    rng = np.random.default_rng(23)
    pix = synth_image(rng, 90, 130)
    Image.fromarray(pix).save(tmp_path / "w.png")
    mine = [files[0], files[3], str(tmp_path / "w.png"), files[5]]
    ref = [decoded[0], decoded[3], pix, decoded[5]]
    pipe = _pipe(mine, 4, decoder="image_random_crop", random_area=[0.1, 0.9], seed=4321, outputs="both")
    for it in range(7):
        out, enc = pipe.run()
        anchors, crops = O.rrc_batch(4321, it, [r.shape[:2] for r in ref], area=(0.1, 0.9))
        for i, r in enumerate(ref):
            (y0, x0), (h, w) = anchors[i], crops[i]
            assert np.array_equal(out[i].as_cpu(), r[y0:y0 + h, x0:x0 + w]), (it, i)
    assert all(enc.at(i).size == 0 for i in range(4)), [enc.at(i).size for i in range(4)]
    assert "jpeg_idct" not in pipe.executed_kernels()        # (no host-decoded coefficients any more: everything JPEG is a device decode)
