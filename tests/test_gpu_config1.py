"""BASELINE.json configs[1] itself, checked against the oracle: the 256 x 3 x 224 x 224 fp16 batch bench.py times
(same synthetic data set, seed 1234, RandomResizedCrop seed 1234 / CoinFlip seed 1235 like bench.HotPath) produced
by dali_amd.Pipeline must EQUAL the oracle composition decode -> RandomResizedCrop -> CropMirrorNormalize bit for
bit; and the committed golden JPEG fixtures (pinned against libjpeg-turbo, tests/golden/make_golden.py) go through
the HIP decode path and must reproduce the recorded sha256."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

MEAN = [0.485 * 255, 0.456 * 255, 0.406 * 255]
STD = [0.229 * 255, 0.224 * 255, 0.225 * 255]
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_configs1_b256_batch_equals_oracle(tmp_path):
    import bench
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    from dali_amd.testing import synth_dataset
    B = 256
    enc = synth_dataset(0, B, seed=1234, workers=4)     # = the first batch of bench.py's data set
    bench.write_dataset(str(tmp_path), enc)
    # readers.file order: class directories sorted, files sorted inside
    order = sorted(range(B), key=lambda g: (g % 10, g))
    pipe = Pipeline(batch_size=B, num_threads=8, device_id=0, seed=1234, prefetch_queue_depth=2)
    with pipe:
        jpegs, labels = fn.readers.file(file_root=str(tmp_path), name="Reader")
        images = fn.decoders.image(jpegs, device="mixed", output_type=types.RGB)
        crops = fn.random_resized_crop(images, size=[224, 224], seed=1234)
        out = fn.crop_mirror_normalize(crops, dtype=types.FLOAT16, output_layout="CHW", mean=MEAN, std=STD,
                                       mirror=fn.random.coin_flip(probability=0.5, seed=1235))
        pipe.set_outputs(out, labels)
    pipe.build()
    mean, inv = O.cmn_norm_args(MEAN, STD)
    for it in range(2):       # the second iteration wraps to the same files with the next crop windows
        data, lab = pipe.run()
        got = data.as_tensor().cpu().numpy()
        assert got.shape == (B, 3, 224, 224) and got.dtype == np.float16
        assert list(lab.as_array().reshape(-1)) == [g % 10 for g in order]
        ref = O.pipeline_batch([enc[g] for g in order], 1234, 1235, it, mean=mean, inv_std=inv, nthreads=8)
        same = got.view(np.uint16) == ref.view(np.uint16)
        bad = np.nonzero(~same.reshape(B, -1).all(1))[0]
        assert same.all(), f"iteration {it}: samples {bad[:8].tolist()} differ from the oracle " \
                           f"({(~same).sum()} of {same.size} elements)"
        assert "jpeg_huffman" in pipe.executed_kernels() and "fused_resample_cmn" in pipe.executed_kernels()


@pytest.mark.parametrize("huffman", ["gpu", "host"])
def test_golden_fixtures_through_the_hip_path(huffman):
    """Every committed golden stream, decoded on the device (GPU Huffman where the stream is eligible - baseline,
    one scan, no restart markers - and host Huffman + IDCT kernel for the rest / for huffman="host"), hashes to the
    value libjpeg-turbo produced when the fixture was made."""
    from dali_amd import backend as B
    gold = json.load(open(os.path.join(GOLDEN, "jpeg_golden.json")))
    names = sorted(gold)
    enc = [open(os.path.join(GOLDEN, n), "rb").read() for n in names]
    views, plan = B.decode_jpeg_batch(enc, device="cuda:0", huffman=huffman)
    if huffman == "gpu":
        assert plan.gpu_eligible.sum() >= 6, "most fixtures must take the GPU entropy decoder"
    for n, v in zip(names, views):
        img = v.cpu().numpy()
        assert list(img.shape) == gold[n]["shape"], n
        assert hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest() == gold[n]["sha256"], n
