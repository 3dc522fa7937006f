"""Parse results of files the mixed decoder has seen before (host/image_cache.h: HeaderCache): from the second epoch of a
shard that is NOT resident the header parse + scan analysis are looked up by file name instead of repeated.  The pixels do
not change; a file that is replaced under the same name is recognised by its size and parsed anew."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import encode_jpeg, synth_image

pytestmark = pytest.mark.gpu


def _pipe(files, batch):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    pipe = Pipeline(batch_size=batch, num_threads=3, device_id=0, prefetch_queue_depth=1, seed=5)
    with pipe:
        enc, _ = fn.readers.file(files=files)
        pipe.set_outputs(fn.decoders.image(enc, device="mixed"))
    return pipe


def test_later_epochs_and_replaced_files(tmp_path):
    rng = np.random.default_rng(61)
    files, ref = [], []
    kinds = [dict(subsampling="4:2:0"), dict(subsampling="4:4:4", optimize=True), dict(subsampling="4:2:2"),
             dict(subsampling="4:2:0", progressive=True), dict(subsampling="4:2:0", restart_marker_blocks=4)]
    for i, (h, w) in enumerate([(120, 160), (200, 150), (97, 131), (64, 48), (75, 211), (180, 180)]):
        data = encode_jpeg(synth_image(rng, h, w), 85, **kinds[i % len(kinds)])
        p = tmp_path / f"h{i}.jpg"
        p.write_bytes(data)
        files.append(str(p))
        ref.append(O.jpeg_decode_rgb(data))
    pipe = _pipe(files, 3)
    for it in range(6):                       # three epochs of two iterations
        (img,) = pipe.run()
        for i in range(3):
            assert np.array_equal(img[i].as_cpu(), ref[(3 * it + i) % 6]), (it, i)
    # the same names, other pictures (other sizes on disk): a second pipeline of the process must not decode them with the
    # geometry it remembers
    ref2 = []
    for i, (h, w) in enumerate([(50, 70), (300, 20), (33, 33), (128, 256), (8, 8), (90, 91)]):
        data = encode_jpeg(synth_image(rng, h, w), 70, **kinds[(i + 2) % len(kinds)])
        assert len(data) != os.path.getsize(files[i])
        open(files[i], "wb").write(data)
        ref2.append(O.jpeg_decode_rgb(data))
    pipe2 = _pipe(files, 6)
    for it in range(2):
        (img,) = pipe2.run()
        for i in range(6):
            assert np.array_equal(img[i].as_cpu(), ref2[i]), (it, i)


def test_a_file_rewritten_under_the_same_name_AND_size_is_parsed_anew(tmp_path):
    """the entry is keyed by path + size; a data set regenerated in place can keep both.  An entry only answers
    for the header bytes it was made from (a hash of everything in front of the entropy-coded segment)."""
    rng = np.random.default_rng(62)
    a = encode_jpeg(synth_image(rng, 120, 160), 85, subsampling="4:2:0")
    b = encode_jpeg(synth_image(rng, 64, 200), 60, subsampling="4:4:4", optimize=True)     # other geometry, other tables
    size = max(len(a), len(b)) + 16
    a, b = a + bytes(size - len(a)), b + bytes(size - len(b))      # same size on disk (bytes behind EOI are not scan data)
    p = tmp_path / "same.jpg"
    p.write_bytes(a)
    pipe = _pipe([str(p)], 1)
    for _ in range(3):
        (img,) = pipe.run()
        assert np.array_equal(img[0].as_cpu(), O.jpeg_decode_rgb(a))
    del pipe
    p.write_bytes(b)
    assert os.path.getsize(p) == size
    pipe2 = _pipe([str(p)], 1)
    for _ in range(3):
        (img,) = pipe2.run()
        assert np.array_equal(img[0].as_cpu(), O.jpeg_decode_rgb(b))
