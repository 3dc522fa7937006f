"""readers.tfrecord / readers.mxnet / readers.webdataset on files written here (TFRecord + tf.Example by hand, RecordIO
incl. a multi-segment record, tar archives via `tarfile`): record contents, shapes and types, the index files, the
shared shard / shuffle / padding semantics, feeding decoders.image.
(reference: tfrecord_parser.h:40-196, recordio_parser.h:31-186, webdataset_loader.cc)"""
import io
import os
import struct
import tarfile

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import encode_jpeg, synth_image


# ------------------------------------------------------------------ writers
def _varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 127
        v >>= 7
        out.append(b | (128 if v else 0))
        if not v:
            return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _feature(value):
    if isinstance(value, bytes):
        return _ld(1, _ld(1, value))                                          # BytesList
    a = np.asarray(value)
    if a.dtype.kind == "f":
        return _ld(2, _ld(1, a.astype("<f4").tobytes()))                      # FloatList, packed
    return _ld(3, _ld(1, b"".join(_varint(int(v)) for v in a.reshape(-1))))   # Int64List, packed


def _example(features):
    entries = b"".join(_ld(1, _ld(1, k.encode()) + _ld(2, _feature(v))) for k, v in features.items())
    return _ld(1, entries)


def _write_tfrecord(path, examples):
    index = []
    with open(path, "wb") as f:
        for ex in examples:
            rec = struct.pack("<Q", len(ex)) + b"\0\0\0\0" + ex + b"\0\0\0\0"    # CRCs are not checked by the reader
            index.append((f.tell(), len(rec)))
            f.write(rec)
    with open(path + ".idx", "w") as f:
        f.writelines(f"{o} {s}\n" for o, s in index)


MAGIC = struct.pack("<I", 0xced7230a)


def _recordio_record(payload):
    """MXNet's writer: a payload containing the magic number is split there into segments (cflag 1, 2.., 3)."""
    parts = payload.split(MAGIC)
    out = b""
    for k, part in enumerate(parts):
        cflag = 0 if len(parts) == 1 else (1 if k == 0 else (3 if k == len(parts) - 1 else 2))
        out += MAGIC + struct.pack("<I", (cflag << 29) | len(part)) + part + b"\0" * (-len(part) % 4)
    return out


def _write_recordio(path, items):
    with open(path, "wb") as f, open(path + ".idx", "w") as idx:
        for k, (labels, image) in enumerate(items):
            if len(labels) == 1:
                hdr = struct.pack("<IfQQ", 0, labels[0], k, 0)
            else:
                hdr = struct.pack("<IfQQ", len(labels), 0.0, k, 0) + np.asarray(labels, "<f4").tobytes()
            idx.write(f"{k}\t{f.tell()}\n")
            f.write(_recordio_record(hdr + image))


@pytest.fixture(scope="module")
def images():
    rng = np.random.default_rng(31)
    return [encode_jpeg(synth_image(rng, 40 + 3 * i, 56 + i), 85) for i in range(11)]


def _pipe(bs, build, **kw):
    from dali_amd.pipeline import Pipeline
    pipe = Pipeline(batch_size=bs, num_threads=3, device_id=None, seed=7, prefetch_queue_depth=1, **kw)
    with pipe:
        outs = build()
        pipe.set_outputs(*(outs if isinstance(outs, (list, tuple)) else [outs]))
    pipe.build()
    return pipe


# ------------------------------------------------------------------ TFRecord
def test_tfrecord_reader(tmp_path, images):
    from dali_amd import fn, tfrecord as tfrec
    a, b = str(tmp_path / "a.tfrecord"), str(tmp_path / "b.tfrecord")
    feats = lambda i: {"image/encoded": images[i], "image/class/label": [i * 10 - 5],          # noqa: E731
                       "bbox": np.arange(4 * (i % 3), dtype=np.float32), "shape": [40 + 3 * i, 56 + i, 3]}
    examples = [_example(feats(i)) for i in range(11)]
    examples[4] = _example({k: v for k, v in feats(4).items() if k != "bbox"})     # a record without the feature
    _write_tfrecord(a, examples[:6])
    _write_tfrecord(b, examples[6:])

    def build():
        out = fn.readers.tfrecord(path=[a, b], index_path=[a + ".idx", b + ".idx"], name="Reader", features={
            "image/encoded": tfrec.FixedLenFeature((), tfrec.string, ""),
            "image/class/label": tfrec.FixedLenFeature([1], tfrec.int64, -1),
            "bbox": tfrec.VarLenFeature([4], tfrec.float32, 0.0),
            "shape": tfrec.FixedLenFeature([3], tfrec.int64, 0)})
        return out["image/encoded"], out["image/class/label"], out["bbox"], out["shape"]
    pipe = _pipe(4, build)
    assert pipe.reader_meta("Reader")["epoch_size"] == 11
    seen = []
    for it in range(3):
        enc, lab, bbox, shp = pipe.run()
        for i in range(4):
            k = (it * 4 + i) % 11
            seen.append(k)
            assert enc.at(i).tobytes() == images[k]
            assert lab.at(i).dtype == np.int64 and lab.at(i).tolist() == [k * 10 - 5]
            assert shp.at(i).tolist() == [40 + 3 * k, 56 + k, 3]
            if k == 4:
                assert bbox.at(i).size == 0
            else:
                assert bbox.at(i).dtype == np.float32 and bbox.at(i).shape == (k % 3, 4)
                assert np.array_equal(bbox.at(i).reshape(-1), np.arange(4 * (k % 3), dtype=np.float32))
    assert seen[:11] == list(range(11))
    with pytest.raises(RuntimeError, match="index files needs to match"):
        _pipe(2, lambda: [fn.readers.tfrecord(path=[a, b], index_path=[a + ".idx"],
                                              features={"x": tfrec.FixedLenFeature([], tfrec.int64, 0)})["x"]])


def test_tfrecord_sharding_shuffle_and_decode(tmp_path, images):
    from dali_amd import fn, tfrecord as tfrec
    a = str(tmp_path / "s.tfrecord")
    _write_tfrecord(a, [_example({"img": images[i], "label": [i]}) for i in range(11)])

    def build(**kw):
        def f():
            out = fn.readers.tfrecord(path=a, index_path=a + ".idx", features={
                "img": tfrec.FixedLenFeature((), tfrec.string, ""), "label": tfrec.FixedLenFeature([1], tfrec.int64, -1)}, **kw)
            return fn.decoders.image(out["img"], device="cpu"), out["label"]
        return f
    # two shards cover the data set between them (loader.cc:78-87: shard k starts at floor(11 k / 2))
    got = []
    for shard in (0, 1):
        pipe = _pipe(6 if shard else 5, build(shard_id=shard, num_shards=2, stick_to_shard=True))
        img, lab = pipe.run()
        labels = [int(lab.at(i)[0]) for i in range(len(lab))]
        got += labels
        for i, k in enumerate(labels):
            assert np.array_equal(img.at(i), O.jpeg_decode_rgb(images[k]))
    assert got == list(range(11))
    # shuffled: a permutation per epoch, reproducible
    p1, p2 = _pipe(11, build(random_shuffle=True, initial_fill=5)), _pipe(11, build(random_shuffle=True, initial_fill=5))
    l1 = [int(x[0]) for x in p1.run()[1].as_array()]
    assert sorted(l1) == list(range(11)) and l1 != list(range(11))
    assert l1 == [int(x[0]) for x in p2.run()[1].as_array()]


# ------------------------------------------------------------------ RecordIO
def test_mxnet_recordio_reader(tmp_path, images):
    from dali_amd import fn
    path = str(tmp_path / "train.rec")
    tricky = images[3][:100] + MAGIC + images[3][100:200] + MAGIC + images[3][200:]     # payload holding the magic number
    items = [([float(i)], images[i]) for i in range(5)] + [([1.5, 2.5, 3.5], images[5]), ([7.0], tricky)]
    _write_recordio(path, items)
    pipe = _pipe(7, lambda: fn.readers.mxnet(path=[path], index_path=[path + ".idx"], name="R"))
    assert pipe.reader_meta("R")["epoch_size"] == 7
    img, lab = pipe.run()
    for i, (labels, image) in enumerate(items):
        assert img.at(i).tobytes() == image, i
        assert lab.at(i).dtype == np.float32 and lab.at(i).tolist() == labels
    with pytest.raises(RuntimeError, match="single index file"):
        _pipe(1, lambda: fn.readers.mxnet(path=[path], index_path=[path + ".idx", path + ".idx"]))


# ------------------------------------------------------------------ WebDataset
def _write_tar(path, entries, index=None):
    with tarfile.open(path, "w", format=tarfile.GNU_FORMAT) as tf:
        for name, data in entries:
            ti = tarfile.TarInfo(name)
            ti.size = len(data)
            tf.addfile(ti, io.BytesIO(data))
    if index:
        with tarfile.open(path) as tf:
            members = [m for m in tf.getmembers() if m.isfile()]
        samples, cur = [], None
        for m in members:
            base, ext = m.name.split(".", 1)
            if base != cur:
                samples.append([])
                cur = base
            samples[-1].append(f"{ext} {m.offset_data} {m.size} {m.name}")
        with open(index, "w") as f:
            f.write(f"v1.2 {len(samples)}\n")
            f.writelines(" ".join(s) + "\n" for s in samples)


@pytest.mark.parametrize("with_index", [False, True])
def test_webdataset_reader(tmp_path, images, with_index):
    from dali_amd import fn, types
    t0, t1 = str(tmp_path / "shard-0.tar"), str(tmp_path / "shard-1.tar")
    long_dir = "d" * 120    # forces a GNU long-name entry
    e0 = [("000.jpg", images[0]), ("000.cls", b"0"), ("000.extra.txt", b"ignored"),
          ("001.JPG", images[1]), ("001.cls", b"1"),
          ("002.png", images[2]), ("002.cls", b"2"),
          (".hidden.jpg", b"x"), ("003.cls", b"3")]                        # 003 has no image
    e1 = [(f"{long_dir}/004.jpg", images[4]), (f"{long_dir}/004.cls", b"4"),
          ("005.jpg", images[5]), ("005.cls", np.arange(3, dtype=np.int32).tobytes())]
    _write_tar(t0, e0, t0 + ".idx" if with_index else None)
    _write_tar(t1, e1, t1 + ".idx" if with_index else None)
    kw = dict(index_paths=[t0 + ".idx", t1 + ".idx"]) if with_index else {}

    def build(**more):
        return lambda: fn.readers.webdataset(paths=[t0, t1], ext=["jpg;png", "cls"], name="W", **kw, **more)
    pipe = _pipe(6, build(case_sensitive_extensions=False))
    assert pipe.reader_meta("W")["epoch_size"] == 6
    img, cls = pipe.run()
    expect = [(images[0], b"0"), (images[1], b"1"), (images[2], b"2"), (b"", b"3"), (images[4], b"4"),
              (images[5], np.arange(3, dtype=np.int32).tobytes())]
    for i, (im, c) in enumerate(expect):
        assert img.at(i).tobytes() == im and cls.at(i).tobytes() == c, i
    # case-sensitive: 001.JPG is not a "jpg"; skip drops the incomplete samples
    pipe = _pipe(4, build(missing_component_behavior="skip"))
    assert pipe.reader_meta("W")["epoch_size"] == 4
    img, cls = pipe.run()
    assert [c.tobytes() for c in (cls.at(i) for i in range(4))] == [b"0", b"2", b"4", np.arange(3, dtype=np.int32).tobytes()]
    with pytest.raises(RuntimeError, match="Underful sample"):
        _pipe(1, build(missing_component_behavior="error"))
    # typed output: the class component of the last sample as int32
    pipe = _pipe(1, lambda: fn.readers.webdataset(paths=[t1], ext=["cls"], dtypes=[types.UINT8], **({"index_paths": [t1 + ".idx"]} if with_index else {})))
    (only,) = pipe.run()
    assert only.at(0).tobytes() == b"4"


def test_webdataset_component_fills_every_output_that_lists_it(tmp_path, images):
    """ext=['jpg', 'jpg;png']: the reference maps an extension to ALL outputs that name it
    (webdataset_loader.cc:413-460), and a single-output reader returns a DataNode, not a list
    (ops/__init__.py:510-514)."""
    from dali_amd import fn
    from dali_amd.data_node import DataNode
    t = str(tmp_path / "s.tar")
    _write_tar(t, [("0.jpg", images[0]), ("1.png", images[1])], None)
    pipe = _pipe(2, lambda: fn.readers.webdataset(paths=[t], ext=["jpg", "jpg;png"], missing_component_behavior="empty"))
    a, b = pipe.run()
    assert a.at(0).tobytes() == images[0] and b.at(0).tobytes() == images[0]
    assert a.at(1).size == 0 and b.at(1).tobytes() == images[1]
    from dali_amd.pipeline import Pipeline
    with Pipeline(batch_size=1, num_threads=1, device_id=None, seed=1):
        assert isinstance(fn.readers.webdataset(paths=[t], ext=["jpg"]), DataNode)


@pytest.mark.parametrize("size_field", [
    b"\xff" * 10 + b"\xfe\x00",        # base-256, all ones: -512 as a signed 64-bit value (used to loop forever)
    b"\xff" * 12,                        # -1
    b"\x80" + b"\x00" * 3 + b"\x7f" + b"\xff" * 7,   # 2^63 - 1
    b"\x80" + b"\x01" + b"\x00" * 10,    # does not fit 64 bits
])
def test_webdataset_rejects_malformed_tar_sizes(tmp_path, images, size_field):
    from dali_amd import fn
    t = str(tmp_path / "bad.tar")
    _write_tar(t, [("0.jpg", images[0]), ("1.jpg", images[1])], None)
    raw = bytearray(open(t, "rb").read())
    raw[124:136] = size_field
    raw[148:156] = b" " * 8
    raw[148:156] = ("%06o\0 " % sum(raw[:512])).encode()
    open(t, "wb").write(bytes(raw))
    with pytest.raises(RuntimeError, match="Malformed tar archive"):
        _pipe(1, lambda: fn.readers.webdataset(paths=[t], ext=["jpg"]))


def test_reader_indexes_near_int64_max_are_refused(tmp_path, images):
    from dali_amd import fn, tfrecord as tfrec
    p = str(tmp_path / "a.tfrecord")
    _write_tfrecord(p, [_example({"image": images[0]})])
    open(p + ".idx", "w").write(f"{2**63 - 2} {2**63 - 2}\n")
    with pytest.raises(RuntimeError, match="does not describe"):
        _pipe(1, lambda: list(fn.readers.tfrecord(path=p, index_path=p + ".idx",
                                                  features={"image": tfrec.FixedLenFeature((), tfrec.string, "")}).values()))
