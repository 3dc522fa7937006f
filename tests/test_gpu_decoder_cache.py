"""GPU: the decoded-image cache of decoders.image (cache_size / cache_type / cache_threshold / cache_debug).
A hit must give exactly the pixels a decode gives, with no decode kernel launched; the entries are handed out in place
(stable device addresses), also while the write of an entry is still in flight on another stream."""
import gc
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import encode_jpeg, synth_image

pytestmark = pytest.mark.gpu


def _colour_stage_ran(kernels):
    """Upsampling + colour conversion of a decode: the colour launch, or (round 4) inside the entropy decoder's block kernel."""
    return "jpeg_color" in kernels or "jpeg_huffman_rgb" in kernels

SIZES = [(120, 160), (200, 150), (97, 131), (240, 320), (64, 48), (333, 500), (180, 180), (75, 211)]


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    root = tmp_path_factory.mktemp("cache")
    rng = np.random.default_rng(5)
    out = []
    for i, hw in enumerate(SIZES):
        kw = dict(subsampling="4:2:0")
        if i == 3:
            kw["progressive"] = True          # host entropy decoder
        p = root / f"img{i}.jpg"
        p.write_bytes(encode_jpeg(synth_image(rng, *hw), 85, **kw))
        out.append(str(p))
    return out


@pytest.fixture(scope="module")
def decoded(files):
    return [O.jpeg_decode_rgb(open(f, "rb").read()) for f in files]


def _pipe(files, batch, **decoder_kw):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    pipe = Pipeline(batch_size=batch, num_threads=3, device_id=0, prefetch_queue_depth=2)
    with pipe:
        enc, _ = fn.readers.file(files=files)
        pipe.set_outputs(fn.decoders.image(enc, device="mixed", **decoder_kw))
    return pipe


@pytest.fixture(autouse=True)
def _collect():
    """One cache per device: the pipelines (and with them the cache) of the previous test must be gone."""
    gc.collect()
    yield
    gc.collect()


def test_threshold_cache_serves_the_second_epoch_in_place(files, decoded):
    pipe = _pipe(files, 4, cache_size=16, cache_type="threshold")
    ptrs = {}
    for it in range(8):                                   # 4 epochs of 2 iterations
        (out,) = pipe.run()
        kernels = pipe.executed_kernels()
        if it < 2:
            assert "jpeg_huffman" in kernels and _colour_stage_ran(kernels)
        else:
            assert kernels == [], (it, kernels)           # every sample is a cache hit: nothing to launch
        for j in range(4):
            i = (it % 2) * 4 + j
            assert np.array_equal(out[j].as_cpu(), decoded[i]), (it, i)
            ptrs.setdefault(i, set()).add(out[j]._ptr)
    assert all(len(p) == 1 for p in ptrs.values()), "a cached image is decoded into its slot and handed out in place"


def test_hits_on_entries_still_being_written(files, decoded):
    """Dataset == one batch: iteration k+1 (other ring slot, other stream) reads what iteration k is still writing."""
    pipe = _pipe(files[:4], 4, cache_size=8, cache_type="threshold")
    for it in range(6):
        (out,) = pipe.run()
        for j in range(4):
            assert np.array_equal(out[j].as_cpu(), decoded[j]), (it, j)


def test_threshold_keeps_only_large_images_and_mixes_hits_with_decodes(files, decoded):
    thr = 100_000                                         # bytes of H*W*3
    big = {i for i, hw in enumerate(SIZES) if hw[0] * hw[1] * 3 >= thr}
    assert 0 < len(big) < len(SIZES)
    log = os.path.join(os.path.dirname(files[0]), "stats.log")
    os.environ["DALI_LOG_FILE"] = log
    try:
        pipe = _pipe(files, 8, cache_size=16, cache_type="threshold", cache_threshold=thr, cache_debug=True)
        for it in range(3):
            (out,) = pipe.run()
            assert _colour_stage_ran(pipe.executed_kernels())   # the small images are decoded every time
            for i in range(8):
                assert np.array_equal(out[i].as_cpu(), decoded[i]), (it, i)
        del out, pipe
        gc.collect()                                      # the cache goes with its last pipeline and prints its report
    finally:
        del os.environ["DALI_LOG_FILE"]
    text = open(log).read()
    assert "CACHE STATS" in text and f"images_cached: {len(big)}" in text and f"images_seen: {len(SIZES)}" in text
    for i in range(8):
        line = [ln for ln in text.splitlines() if files[i] in ln][0]
        assert f"is_cached[{int(i in big)}]" in line
        # the prefetching executor issued 3 + 2 iterations: one decode for a kept image, all of them otherwise
        assert ("decodes[1]" in line) == (i in big), line


def test_largest_policy_warms_up_in_two_epochs(files, decoded):
    # stored size of an image: rows padded to 16 bytes, slots to 256 bytes
    stored = sorted((((h * ((w * 3 + 15) // 16 * 16) + 255) // 256 * 256, i) for i, (h, w) in enumerate(SIZES)),
                    reverse=True)
    keep, total = set(), 0
    for size, i in stored:                                # 1 MB holds all but one (97x131) of the eight images
        if total + size <= 1 << 20:
            keep.add(i)
            total += size
    assert keep == {0, 1, 3, 4, 5, 6, 7}
    log = os.path.join(os.path.dirname(files[0]), "largest.log")
    os.environ["DALI_LOG_FILE"] = log
    try:
        pipe = _pipe(files, 8, cache_size=1, cache_type="largest", cache_debug=True)
        ptrs = []
        for it in range(5):
            (out,) = pipe.run()
            for i in range(8):
                assert np.array_equal(out[i].as_cpu(), decoded[i]), (it, i)
            ptrs.append([out[i]._ptr for i in range(8)])
        del out, pipe
        gc.collect()
    finally:
        del os.environ["DALI_LOG_FILE"]
    text = open(log).read()
    assert f"images_cached: {len(keep)}" in text and "is_cache_full: 1" in text
    for i in range(8):
        line = [ln for ln in text.splitlines() if files[i] in ln][0]
        # epoch 1 ranks, epoch 2 stores, from epoch 3 on the kept images come from the cache
        assert f"is_cached[{int(i in keep)}]" in line, line
        assert ("decodes[2]" in line) == (i in keep), line
        if i in keep:
            assert ptrs[1][i] == ptrs[2][i] == ptrs[3][i] == ptrs[4][i], "decoded into the slot, handed out in place"


def test_cached_batches_feed_the_fused_resample_kernel(files, decoded):
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    mean, std = [0.485 * 255, 0.456 * 255, 0.406 * 255], [0.229 * 255, 0.224 * 255, 0.225 * 255]

    def build(**kw):
        pipe = Pipeline(batch_size=8, num_threads=3, device_id=0, prefetch_queue_depth=2, seed=3)
        with pipe:
            enc, _ = fn.readers.file(files=files)
            img = fn.decoders.image(enc, device="mixed", **kw)
            img = fn.random_resized_crop(img, size=[64, 64], seed=11)
            pipe.set_outputs(fn.crop_mirror_normalize(img, dtype=types.FLOAT16, mean=mean, std=std))
        return pipe
    plain, cached = build(), build(cache_size=16, cache_type="threshold")
    for it in range(4):
        (a,), (b,) = plain.run(), cached.run()
        assert _colour_stage_ran(cached.executed_kernels()) == (it == 0)
        assert "fused_resample_cmn" in cached.executed_kernels()
        assert np.array_equal(a.as_tensor().cpu().numpy().view(np.uint16), b.as_tensor().cpu().numpy().view(np.uint16)), it


def test_one_cache_per_device_with_one_set_of_parameters(files):
    with pytest.raises(RuntimeError, match="unexpected cache policy"):
        _pipe(files, 4, cache_size=4, cache_type="lru").build()
    first = _pipe(files, 4, cache_size=4, cache_type="threshold")
    first.build()
    same = _pipe(files, 4, cache_size=4, cache_type="threshold")     # shares the cache of `first`
    same.build()
    (out,) = first.run()
    (out2,) = same.run()
    assert same.executed_kernels() == [] and out2[0]._ptr == out[0]._ptr
    other = _pipe(files, 4, cache_size=8, cache_type="threshold")
    with pytest.raises(RuntimeError, match="already initialized with other parameters"):
        other.build()
