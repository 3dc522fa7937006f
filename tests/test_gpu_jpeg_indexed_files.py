"""GPU: a data set indexed offline (tools/jpeg2idx.py) read through readers.file(index_path=...) and decoded by the mixed
decoders FROM the containers' index entries - `jpeg_huffman_indexed` in the very first iteration of a fresh pipeline, where
cache_type="indexed" only gets there in the second epoch.  Bit for bit the oracle's pixels: whole images, region-of-interest
decoders, batches that mix containers with files that have none (progressive, PNG), the staged, the direct and the
device-fetched (zero-copy reader) transfer, and with the encoded cache on top (a container stays resident with its entry)."""
import gc
import os
import sys

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import encode_jpeg, synth_image

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

SIZES = [(120, 160), (200, 150), (97, 131), (240, 320), (64, 48), (333, 500), (180, 180), (75, 211)]


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    import jpeg2idx
    root, idx = tmp_path_factory.mktemp("didx_data"), tmp_path_factory.mktemp("didx_index")
    rng = np.random.default_rng(41)
    files, want = [], []
    for i, hw in enumerate(SIZES):
        kw = dict(subsampling=["4:2:0", "4:4:4", "4:2:2"][i % 3])
        if i == 3:
            kw["progressive"] = True                       # gets no container
        if i == 5:
            kw["optimize"] = True
        enc = encode_jpeg(synth_image(rng, *hw), 85, **kw)
        (root / f"img{i}.jpg").write_bytes(enc)
        files.append(str(root / f"img{i}.jpg"))
        want.append(O.jpeg_decode_rgb(enc))
    made, skipped = jpeg2idx.index_tree(str(root), str(idx), quiet=True)
    assert (made, skipped) == (7, 1)
    return str(root), str(idx), files, want


@pytest.fixture(autouse=True)
def _collect():
    gc.collect()
    yield
    gc.collect()


def _pipe(root, idx, files, batch, decoder="image", outputs="image", reader_kw=None, **decoder_kw):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    pipe = Pipeline(batch_size=batch, num_threads=3, device_id=0, prefetch_queue_depth=2, seed=11)
    with pipe:
        enc, _ = fn.readers.file(file_root=root, files=[os.path.basename(f) for f in files], index_path=idx, **(reader_kw or {}))
        img = getattr(fn.decoders, decoder)(enc, device="mixed", **decoder_kw)
        pipe.set_outputs(*((img, enc) if outputs == "both" else (img,)))
    return pipe


def test_first_iteration_of_a_fresh_pipeline_decodes_from_the_index(dataset):
    root, idx, files, want = dataset
    base = [f for k, f in enumerate(files) if k != 3]               # only files with a container: the direct transfer
    ref = [w for k, w in enumerate(want) if k != 3]
    pipe = _pipe(root, idx, base, len(base), outputs="both")
    img, enc = pipe.run()
    assert "jpeg_huffman_indexed" in pipe.executed_kernels()
    for i in range(len(base)):
        assert bytes(np.asarray(enc.at(i))[:4]) == b"DAJX"
        assert np.array_equal(img[i].as_cpu(), ref[i]), i
    for it in range(3):
        (img, _) = pipe.run()
        for i in range(len(base)):
            assert np.array_equal(img[i].as_cpu(), ref[i]), (it, i)


def test_batches_that_mix_containers_progressive_files_and_windows(dataset):
    root, idx, files, want = dataset
    pipe = _pipe(root, idx, files, 4)                                # the progressive file has no container: staged transfer
    for it in range(6):
        (img,) = pipe.run()
        for i in range(4):
            k = (4 * it + i) % len(files)
            assert np.array_equal(img[i].as_cpu(), want[k]), (it, i, k)
    assert "jpeg_huffman_indexed" in pipe.executed_kernels()
    pipe = _pipe(root, idx, files, len(files), decoder="image_random_crop", random_area=[0.2, 0.8], seed=1234)
    for it in range(4):
        (out,) = pipe.run()
        anchors, crops = O.rrc_batch(1234, it, [r.shape[:2] for r in want], area=(0.2, 0.8))
        for i, r in enumerate(want):
            (y0, x0), (h, w) = anchors[i], crops[i]
            assert np.array_equal(out[i].as_cpu(), r[y0:y0 + h, x0:x0 + w]), (it, i)


@pytest.mark.parametrize("cache_type", ["encoded", "indexed"])
def test_containers_become_resident_with_their_entry(dataset, cache_type):
    root, idx, files, want = dataset
    pipe = _pipe(root, idx, files, 4, outputs="both", reader_kw=dict(skip_cached_images=True), cache_size=64, cache_type=cache_type)
    for it in range(10):
        img, enc = pipe.run()
        for i in range(4):
            k = (4 * it + i) % len(files)
            assert np.array_equal(img[i].as_cpu(), want[k]), (it, i, k)
            if it >= 4:
                assert enc.at(i).size == 0, (it, k)                 # resident: nothing is read any more
        if it >= 4:
            assert "jpeg_huffman_indexed" in pipe.executed_kernels()   # ... and still decoded from the entry, whatever cache_type


def test_device_fetch_of_containers(dataset, monkeypatch):
    monkeypatch.setenv("DALI_AMD_READER_ZERO_COPY", "1")
    root, idx, files, want = dataset
    base = [f for k, f in enumerate(files) if k != 3]
    ref = [w for k, w in enumerate(want) if k != 3]
    pipe = _pipe(root, idx, base, len(base))
    seen = False
    for it in range(6):
        (img,) = pipe.run()
        for i in range(len(base)):
            assert np.array_equal(img[i].as_cpu(), ref[i]), (it, i)
        seen = seen or "gather_encoded" in pipe.executed_kernels()
    assert seen and "jpeg_huffman_indexed" in pipe.executed_kernels()


def test_headline_graph_from_an_indexed_data_set_equals_oracle(tmp_path):
    """bench.py's e2e leg with index_path at batch 256: decode -> RandomResizedCrop -> CropMirrorNormalize == the oracle."""
    import bench
    import jpeg2idx
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    from dali_amd.testing import synth_dataset
    batch = 256
    enc = synth_dataset(0, batch, seed=1234, workers=8)
    root, idx = str(tmp_path / "data"), str(tmp_path / "index")
    bench.write_dataset(root, enc)
    made, skipped = jpeg2idx.index_tree(root, idx, workers=8, quiet=True)
    assert made == batch and skipped == 0
    order = sorted(range(batch), key=lambda g: (g % 10, g))
    mean_a, std_a = [0.485 * 255, 0.456 * 255, 0.406 * 255], [0.229 * 255, 0.224 * 255, 0.225 * 255]
    pipe = Pipeline(batch_size=batch, num_threads=8, device_id=0, seed=1234, prefetch_queue_depth=5)
    with pipe:
        jpegs, labels = fn.readers.file(file_root=root, index_path=idx, name="Reader")
        images = fn.decoders.image(jpegs, device="mixed", output_type=types.RGB)
        crops = fn.random_resized_crop(images, size=[224, 224], seed=1234)
        out = fn.crop_mirror_normalize(crops, dtype=types.FLOAT16, output_layout="CHW", mean=mean_a, std=std_a,
                                       mirror=fn.random.coin_flip(probability=0.5, seed=1235))
        pipe.set_outputs(out, labels)
    mean, inv = O.cmn_norm_args(mean_a, std_a)
    for it in range(3):
        data, lab = pipe.run()
        got = data.as_tensor().cpu().numpy()
        ref = O.pipeline_batch([enc[g] for g in order], 1234, 1235, it, mean=mean, inv_std=inv, nthreads=8)
        same = got.view(np.uint16) == ref.view(np.uint16)
        assert same.all(), f"iteration {it}: samples {np.nonzero(~same.reshape(batch, -1).all(1))[0][:8].tolist()} differ"
    assert "jpeg_huffman_indexed" in pipe.executed_kernels() and "windows_of_the_consumer" in pipe.executed_kernels()
