"""Mutation fuzzing of the container readers (untrusted files): TFRecord + index, RecordIO + index and tar archives
(with and without a webdataset index) written by the helpers of tests/test_container_readers.py are corrupted - bit
flips, truncation, random runs, swapped index lines - and read through CPU pipelines.  Building / running either works or
raises; anything else is a bug.  Meant for the AddressSanitizer build:  tools/asan_fuzz.sh N readers"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tests import test_container_readers as T
from tests.util import encode_jpeg, synth_image


def corrupt(rng, path):
    d = bytearray(open(path, "rb").read())
    if not d:
        return
    kind = rng.integers(0, 5)
    if kind == 0:
        del d[rng.integers(0, len(d)):]
    elif kind == 1:
        for _ in range(rng.integers(1, 6)):
            i = rng.integers(0, len(d)); d[i] ^= 1 << rng.integers(0, 8)
    elif kind == 2:
        i = rng.integers(0, len(d)); n = min(len(d) - i, int(rng.integers(1, 32)))
        d[i:i + n] = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    elif kind == 3:
        i = rng.integers(0, len(d)); n = min(len(d) - i, int(rng.integers(1, 9)))
        d[i:i + n] = bytes([0xFF if rng.integers(0, 2) else 0x00]) * n
    else:
        i = rng.integers(0, len(d)); n = min(len(d) - i, int(rng.integers(1, 600)))
        del d[i:i + n]
    open(path, "wb").write(bytes(d))


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    from dali_amd import fn, tfrecord as tfrec
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(99)
    images = [encode_jpeg(synth_image(rng, 24 + i, 32 + i), 80) for i in range(6)]
    ok = bad = 0
    for it in range(iters):
        with tempfile.TemporaryDirectory() as tmp:
            tfr, rec, tar = os.path.join(tmp, "a.tfrecord"), os.path.join(tmp, "a.rec"), os.path.join(tmp, "a.tar")
            T._write_tfrecord(tfr, [T._example({"image/encoded": images[i], "label": [i], "box": np.arange(4 * (i % 2), dtype=np.float32)})
                                    for i in range(6)])
            T._write_recordio(rec, [([float(i)], images[i]) for i in range(5)] + [([1.0, 2.0], images[5])])
            with_index = bool(it & 1)
            T._write_tar(tar, [(f"{i:03d}.jpg", images[i]) for i in range(6)] + [(f"{i:03d}.cls", str(i).encode()) for i in range(6)],
                         tar + ".idx" if with_index else None)
            victims = [tfr, tfr + ".idx", rec, rec + ".idx", tar] + ([tar + ".idx"] if with_index else [])
            for v in rng.choice(len(victims), size=rng.integers(1, 3), replace=False):
                corrupt(rng, victims[v])
            builders = [
                lambda: list(fn.readers.tfrecord(path=[tfr], index_path=[tfr + ".idx"], features={
                    "image/encoded": tfrec.FixedLenFeature((), tfrec.string, ""), "label": tfrec.FixedLenFeature([1], tfrec.int64, -1),
                    "box": tfrec.VarLenFeature([4], tfrec.float32, 0.0)}).values()),
                lambda: list(fn.readers.mxnet(path=[rec], index_path=[rec + ".idx"])),
                lambda: list(fn.readers.webdataset(paths=[tar], ext=["jpg", "cls"], **({"index_paths": [tar + ".idx"]} if with_index else {}))),
            ]
            for build in builders:
                try:
                    pipe = Pipeline(batch_size=3, num_threads=2, device_id=None, prefetch_queue_depth=1)
                    with pipe:
                        pipe.set_outputs(*build())
                    pipe.build()
                    for _ in range(3):
                        pipe.run()
                    ok += 1
                except (RuntimeError, ValueError):
                    bad += 1
    print(f"reader fuzz: {ok} pipelines ran, {bad} rejected with an error, no crash")


if __name__ == "__main__":
    main()
