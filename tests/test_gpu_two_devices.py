"""Two devices driven from ONE process, the reference's multi-GPU iterator form: DALIGenericIterator takes a list of
pipelines, one per GPU, each reading its own shard (dali/python/nvidia/dali/plugin/pytorch/__init__.py:178-180,
docs/advanced_topics_sharding.rst:4-17; readers: loader/file_label_loader.cc:34-92).  Every gpu test elsewhere uses
device_id=0; this one puts the headline graph on device 0 AND device 1 - per-device encoded-stream caches, per-device
streams and ring slots, set_affinity's NUMA lookup for a second device - and holds both to the oracle bit for bit.
Skipped on a one-GPU box (the development pool); it runs wherever the driver has a multi-GPU node."""
import gc

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

MEAN = [0.485 * 255, 0.456 * 255, 0.406 * 255]
STD = [0.229 * 255, 0.224 * 255, 0.225 * 255]


def _device_count():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs in this process")
def test_iterator_over_two_devices_equals_oracle_per_shard(tmp_path):
    import torch
    import bench
    from dali_amd import _backend
    from dali_amd.plugin.pytorch import DALIGenericIterator
    from dali_amd.testing import synth_dataset
    batch, batches, epochs, depth = 64, 2, 3, 3
    per_shard = batch * batches
    n = 2 * per_shard
    enc = synth_dataset(0, n, seed=1234, workers=4)
    bench.write_dataset(str(tmp_path), enc)
    order = sorted(range(n), key=lambda g: (g % 10, g))          # readers.file: class directories sorted, files sorted inside
    shards = [order[r * per_shard:(r + 1) * per_shard] for r in range(2)]   # contiguous shards (loader.cc:78-87)
    before = [_backend.encoded_cache_stats(d) for d in range(2)]
    pipes = [bench.resident_pipeline(str(tmp_path), batch, dev, depth, 4, shard_id=dev, num_shards=2, cache_mb=64,
                                     crop_seed=1234, flip_seed=1235) for dev in range(2)]
    it = DALIGenericIterator(pipes, ["data", "label"])
    mean, inv = O.cmn_norm_args(MEAN, STD)
    for k in range(batches * epochs):
        out = next(it)
        assert len(out) == 2
        for dev in range(2):
            data, lab = out[dev]["data"], out[dev]["label"]
            assert data.device == torch.device("cuda", dev) and data.dtype == torch.float16
            torch.cuda.synchronize(dev)
            ids = shards[dev][(k % batches) * batch:(k % batches + 1) * batch]
            assert lab.reshape(-1).tolist() == [g % 10 for g in ids]
            ref = O.pipeline_batch([enc[g] for g in ids], 1234, 1235, k, mean=mean, inv_std=inv, nthreads=8)
            got = data.cpu().numpy()
            same = got.view(np.uint16) == ref.view(np.uint16)
            bad = np.nonzero(~same.reshape(batch, -1).all(1))[0]
            assert same.all(), f"device {dev}, iteration {k}: samples {bad[:8].tolist()} differ from the oracle"
    after = [_backend.encoded_cache_stats(d) for d in range(2)]
    for dev in range(2):   # one cache per device: each holds its own shard, and served it from the second epoch on
        assert after[dev]["streams"] - before[dev]["streams"] == per_shard
        assert after[dev]["hits"] - before[dev]["hits"] >= batch * (batches * (epochs - 1) - depth - 1)
        assert "jpeg_huffman" in pipes[dev].executed_kernels()
    del it, pipes
    gc.collect()


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs in this process")
def test_file_fed_pipeline_on_device_one(tmp_path):
    """The from-files path (reader -> H2D -> decode -> fused resample) with device_id=1 only and set_affinity=True: nothing
    of it may touch device 0's streams or caches."""
    import torch
    import bench
    from dali_amd.testing import synth_dataset
    batch = 32
    enc = synth_dataset(0, 2 * batch, seed=1234, workers=4)
    bench.write_dataset(str(tmp_path), enc)
    order = sorted(range(2 * batch), key=lambda g: (g % 10, g))
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    pipe = Pipeline(batch_size=batch, num_threads=4, device_id=1, seed=7, prefetch_queue_depth=2, set_affinity=True)
    with pipe:
        jpegs, labels = fn.readers.file(file_root=str(tmp_path), name="Reader")
        images = fn.decoders.image(jpegs, device="mixed", output_type=types.RGB)
        crops = fn.random_resized_crop(images, size=[224, 224], seed=1234)
        out = fn.crop_mirror_normalize(crops, dtype=types.FLOAT16, output_layout="CHW", mean=MEAN, std=STD,
                                       mirror=fn.random.coin_flip(probability=0.5, seed=1235))
        pipe.set_outputs(out, labels)
    pipe.build()
    mean, inv = O.cmn_norm_args(MEAN, STD)
    for k in range(4):
        data, lab = pipe.run()
        assert data.device_id() == 1
        got = data.as_tensor()
        assert got.device == torch.device("cuda", 1)
        ids = order[(k % 2) * batch:(k % 2 + 1) * batch]
        ref = O.pipeline_batch([enc[g] for g in ids], 1234, 1235, k, mean=mean, inv_std=inv, nthreads=8)
        assert np.array_equal(got.cpu().numpy().view(np.uint16), ref.view(np.uint16)), k
