"""The composition bench.py reports as `value`, held to the oracle: bench.resident_pipeline - decoders.image with
cache_type="encoded" + readers.file(skip_cached_images, stick_to_shard), prefetch_queue_depth 5 on three compute streams,
batch 256 - run for several epochs so that every sample is served from the resident streams on every stream of the
rotation, each iteration bit for bit equal to decode -> RandomResizedCrop -> CropMirrorNormalize of the oracle.  Same at
batch 512, the per-GPU batch of BASELINE.json configs[4].  (Each ingredient has its own test - test_gpu_config1,
test_gpu_encoded_cache, test_gpu_pipeline; this is their product.)"""
import gc
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

MEAN = [0.485 * 255, 0.456 * 255, 0.406 * 255]
STD = [0.229 * 255, 0.224 * 255, 0.225 * 255]


@pytest.fixture(autouse=True)
def _collect():   # the encoded-stream cache of a device lives as long as a pipeline that uses it
    gc.collect()
    yield
    gc.collect()


@pytest.mark.parametrize("batch,batches,epochs,cache_type", [(256, 2, 4, "encoded"), (512, 1, 4, "encoded"), (256, 2, 4, "indexed")])
def test_resident_pipeline_of_the_bench_equals_oracle(tmp_path, batch, batches, epochs, cache_type):
    import bench
    from dali_amd import _backend
    from dali_amd.testing import synth_dataset
    n = batch * batches
    enc = synth_dataset(0, n, seed=1234, workers=4)      # the first images of bench.py's data set
    bench.write_dataset(str(tmp_path), enc)
    order = sorted(range(n), key=lambda g: (g % 10, g))  # readers.file: class directories sorted, files sorted inside
    depth = 5                                            # bench.py --inflight default
    before = _backend.encoded_cache_stats(0)
    pipe = bench.resident_pipeline(str(tmp_path), batch, 0, depth, 8, cache_mb=max(64, int(2.2 * sum(map(len, enc)) / 2**20)),
                                   crop_seed=1234, flip_seed=1235, cache_type=cache_type)
    mean, inv = O.cmn_norm_args(MEAN, STD)
    for it in range(batches * epochs):
        data, lab = pipe.run()
        got = data.as_tensor().cpu().numpy()
        ids = order[(it % batches) * batch:(it % batches + 1) * batch]
        assert got.shape == (batch, 3, 224, 224) and got.dtype == np.float16
        assert list(lab.as_array().reshape(-1)) == [g % 10 for g in ids]
        ref = O.pipeline_batch([enc[g] for g in ids], 1234, 1235, it, mean=mean, inv_std=inv, nthreads=8)
        same = got.view(np.uint16) == ref.view(np.uint16)
        bad = np.nonzero(~same.reshape(batch, -1).all(1))[0]
        assert same.all(), f"iteration {it}: samples {bad[:8].tolist()} differ from the oracle"
    after = _backend.encoded_cache_stats(0)
    # every stream became resident during the first epoch; what was decoded later never came from a file.  (The
    # reader runs `depth` batches ahead, so a few batches of the second epoch may still have been read.)
    assert after["streams"] - before["streams"] == n
    assert after["hits"] - before["hits"] >= batch * (batches * (epochs - 1) - depth - 1)
    assert "jpeg_huffman" in pipe.executed_kernels() and "fused_resample_cmn" in pipe.executed_kernels()
    # the graph-level fusion (the decoder decodes the windows the crop draws) is the default: bench.py's algorithmic bytes
    # and config.roi_decode_fusion follow executed_kernels(), so a fusion that silently disengaged must fail HERE
    if os.environ.get("DALI_AMD_ROI_FUSION", "1") != "0":
        assert "windows_of_the_consumer" in pipe.executed_kernels()
    # cache_type="indexed" (bench.py's resident_indexed leg): the later epochs decoded from the streams' side information
    assert ("jpeg_huffman_indexed" in pipe.executed_kernels()) == (cache_type == "indexed")


def test_fused_equals_unfused_at_the_bench_shape(tmp_path):
    """The headline composition itself - batch 256, prefetch_queue_depth 5 (three compute streams), resident streams - with
    the window decode on and off: the same bits, iteration by iteration, over three epochs (test_gpu_roi_fusion.py holds
    the two to each other at batch 10, depth 2, from files)."""
    import bench
    from dali_amd.testing import synth_dataset
    batch, batches, epochs, depth = 256, 2, 3, 5
    n = batch * batches
    enc = synth_dataset(0, n, seed=1234, workers=4)
    bench.write_dataset(str(tmp_path), enc)
    runs = {}
    for fusion in (True, False):
        pipe = bench.resident_pipeline(str(tmp_path), batch, 0, depth, 8, cache_mb=max(64, int(2 * sum(map(len, enc)) / 2**20)),
                                       crop_seed=1234, flip_seed=1235, roi_fusion=fusion)
        runs[fusion] = [pipe.run()[0].as_tensor().cpu().numpy().view(np.uint16).copy() for _ in range(batches * epochs)]
        assert ("windows_of_the_consumer" in pipe.executed_kernels()) == fusion
        del pipe
        gc.collect()
    for it, (a, b) in enumerate(zip(runs[True], runs[False])):
        bad = np.nonzero((a != b).reshape(batch, -1).any(1))[0]
        assert bad.size == 0, f"iteration {it}: samples {bad[:8].tolist()} differ between the fused and the full decode"
