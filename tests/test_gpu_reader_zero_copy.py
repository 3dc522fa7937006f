"""GPU: readers.file hands out page-locked RESIDENT COPIES of its files from a file's second sighting on (no host copy per
epoch) and the mixed decoder fetches the entropy-coded segments with a device-side copy (`gather_encoded` in
executed_kernels()).  Round 5 handed out the file mappings themselves, registered with the device: truncating a file then
hung the process's GPU queues for minutes.  The resident copies are anonymous memory: the last test truncates the files
under a live reader and the pipeline keeps producing the same batches.  The batches must be those of the copying path, bit for bit: whole images, region-of-interest decodes,
batches that mix JPEGs the device decodes, a progressive one (host decoder, reads the mapping itself) and a PNG."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import encode_jpeg, synth_image

pytestmark = pytest.mark.gpu

SIZES = [(120, 160), (200, 150), (97, 131), (240, 320), (64, 48), (333, 500), (180, 180), (75, 211)]


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    root = tmp_path_factory.mktemp("zero_copy")
    rng = np.random.default_rng(77)
    files, want = [], []
    for i, hw in enumerate(SIZES):
        kw = dict(subsampling=["4:2:0", "4:4:4", "4:2:2"][i % 3])
        if i == 3:
            kw["progressive"] = True
        enc = encode_jpeg(synth_image(rng, *hw), 85, **kw)
        p = root / f"img{i}.jpg"
        p.write_bytes(enc)
        files.append(str(p))
        want.append(O.jpeg_decode_rgb(enc))
    from PIL import Image
    pix = synth_image(rng, 50, 70)
    Image.fromarray(pix).save(root / "extra.png")
    files.append(str(root / "extra.png"))
    want.append(pix)
    return files, want


def _pipe(files, batch, decoder="image", both=False, **kw):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    pipe = Pipeline(batch_size=batch, num_threads=3, device_id=0, prefetch_queue_depth=2, seed=5)
    with pipe:
        enc, _ = fn.readers.file(files=files)
        img = getattr(fn.decoders, decoder)(enc, device="mixed", **kw)
        pipe.set_outputs(*((img, enc) if both else (img,)))
    return pipe


@pytest.fixture(autouse=True)
def _zero_copy_on(monkeypatch):
    # (the default follows the number of CPUs the process may use: on below four)
    monkeypatch.setenv("DALI_AMD_READER_ZERO_COPY", "1")


def test_second_sighting_is_fetched_by_the_device_and_decodes_to_the_same_bits(dataset):
    files, want = dataset
    pipe = _pipe(files, 3, both=True)
    seen = False
    for it in range(12):                                   # four epochs of nine files
        img, enc = pipe.run()
        for i in range(3):
            k = (3 * it + i) % len(files)
            assert np.array_equal(img[i].as_cpu(), want[k]), (it, i, k)
            assert bytes(np.asarray(enc.at(i))) == open(files[k], "rb").read(), (it, i)   # the hand-out IS the file
        if "gather_encoded" in pipe.executed_kernels():
            seen = True
    assert seen, "no batch was fetched out of the resident copies"


def test_region_of_interest_decoders_from_the_mappings(dataset):
    files, want = dataset
    jpegs, ref = files[:3] + files[4:8], want[:3] + want[4:8]
    pipe = _pipe(jpegs, len(jpegs), decoder="image_random_crop", random_area=[0.2, 0.8], seed=1234)
    for it in range(4):
        (out,) = pipe.run()
        anchors, crops = O.rrc_batch(1234, it, [r.shape[:2] for r in ref], area=(0.2, 0.8))
        for i, r in enumerate(ref):
            (y0, x0), (h, w) = anchors[i], crops[i]
            assert np.array_equal(out[i].as_cpu(), r[y0:y0 + h, x0:x0 + w]), (it, i)
    assert "gather_encoded" in pipe.executed_kernels()


def test_switched_off_the_reader_copies_as_before(dataset, monkeypatch):
    files, want = dataset
    monkeypatch.setenv("DALI_AMD_READER_ZERO_COPY", "0")
    pipe = _pipe(files[:3], 3)
    for it in range(4):
        (img,) = pipe.run()
        for i in range(3):
            assert np.array_equal(img[i].as_cpu(), want[i])
    assert "gather_encoded" not in pipe.executed_kernels()


def test_truncating_the_files_under_a_live_zero_copy_reader_neither_hangs_nor_changes_the_batches(dataset, tmp_path):
    """a data set file rewritten in place during training must not be a hung job.  No decoder cache
    here - every epoch the device fetches the reader's resident copies."""
    import shutil
    import time
    files, want = dataset
    mine = []
    for k in (0, 1, 2, 4, 5, 6):
        dst = tmp_path / f"t{k}.jpg"
        shutil.copy(files[k], dst)
        mine.append(str(dst))
    ref = [want[k] for k in (0, 1, 2, 4, 5, 6)]
    pipe = _pipe(mine, 3)
    for it in range(6):                                   # three epochs: every file is resident, the reader is ahead
        (img,) = pipe.run()
    assert "gather_encoded" in pipe.executed_kernels()
    for f in mine[:3]:
        open(f, "wb").close()                             # truncate to nothing
    for f in mine[3:]:
        with open(f, "r+b") as fh:                        # rewrite in place with other bytes of the same length
            n = len(fh.read())
            fh.seek(0)
            fh.write(b"\x00" * n)
    t0 = time.perf_counter()
    for it in range(8):
        (img,) = pipe.run()
        for i in range(3):
            assert np.array_equal(img[i].as_cpu(), ref[(3 * it + i) % 6]), (it, i)
    assert time.perf_counter() - t0 < 30.0
    assert "gather_encoded" in pipe.executed_kernels()
