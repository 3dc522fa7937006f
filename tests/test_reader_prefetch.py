"""readers.file runs ahead of the executor on its own threads (reader_op.h:57-183,386-415: prefetch thread + queue of
`prefetch_queue_depth` batches; loader.h:231-272).  What must not change with the depth: the sample stream, the labels, the
checkpoint semantics (a checkpoint describes what was HANDED OUT, not what was read ahead), error reporting."""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    root = tmp_path_factory.mktemp("ds_prefetch")
    files = []
    k = 0
    rng = np.random.default_rng(7)
    for c, n in (("a", 40), ("b", 33), ("c", 30)):
        os.makedirs(root / c)
        for i in range(n):
            p = root / c / f"img_{i:03d}.jpg"
            # the global index, then a tail of varying length: every sample has its own size and content
            p.write_bytes(np.int32(k).tobytes() + rng.integers(0, 256, int(rng.integers(1, 5000)), dtype=np.uint8).tobytes())
            files.append((str(p), {"a": 0, "b": 1, "c": 2}[c]))
            k += 1
    return str(root), files


def _pipe(root, bs, exec_depth=2, **kw):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    checkpoint = kw.pop("checkpoint", None)
    pipe = Pipeline(batch_size=bs, num_threads=3, device_id=None, seed=11, prefetch_queue_depth=exec_depth,
                    **({"checkpoint": checkpoint} if checkpoint else {}))
    with pipe:
        data, label = fn.readers.file(file_root=root, name="Reader", **kw)
        pipe.set_outputs(data, label)
    pipe.build()
    return pipe


def _take(pipe, iters, files=None):
    out = []
    for _ in range(iters):
        d, l = pipe.run()
        for i in range(len(d)):
            raw = d.at(i).tobytes()
            idx = int(np.frombuffer(raw[:4], np.int32)[0])
            if files is not None:   # the whole file arrived, at the right place, with its label
                assert raw == open(files[idx][0], "rb").read()
                assert int(l.at(i)[0]) == files[idx][1]
            out.append(idx)
    return out


@pytest.mark.parametrize("shuffle", [False, True])
def test_stream_does_not_depend_on_how_far_the_reader_runs_ahead(dataset, shuffle):
    root, files = dataset
    runs = []
    for depth, exec_depth in ((1, 1), (2, 3), (5, 2)):
        pipe = _pipe(root, 16, exec_depth=exec_depth, prefetch_queue_depth=depth, random_shuffle=shuffle, initial_fill=37,
                     pad_last_batch=True)
        runs.append(_take(pipe, 20, files))
    assert runs[0] == runs[1] == runs[2]
    assert sorted(set(runs[0][:103])) == list(range(103))   # one epoch = every file once (103 < 7 * 16: padded)


def test_checkpoint_is_the_position_of_the_consumer_not_of_the_read_ahead(dataset):
    root, _ = dataset
    kw = dict(prefetch_queue_depth=4, random_shuffle=True, initial_fill=29, shard_id=1, num_shards=2)
    pipe = _pipe(root, 8, **kw)
    _take(pipe, 7)
    cpt = pipe.checkpoint()           # the reader threads are four batches further by now
    want = _take(pipe, 9)
    fresh = _pipe(root, 8, checkpoint=cpt, **kw)
    assert _take(fresh, 9) == want
    # ... and restoring into a pipeline that has already run (its read-ahead is dropped)
    pipe2 = _pipe(root, 8, **kw)
    _take(pipe2, 3)
    pipe2.checkpoint()                # (drains the executor's own prefetch: restore wants an idle pipeline)
    pipe2._backend.restore(cpt)
    assert _take(pipe2, 9) == want


def test_descriptor_cache_evicts_without_losing_a_read(dataset, monkeypatch):
    root, files = dataset
    monkeypatch.setenv("DALI_AMD_READER_FD_CAP", "7")    # far fewer descriptors than files: constant evictions
    pipe = _pipe(root, 16, prefetch_queue_depth=3)
    got = _take(pipe, 26, files)                          # four epochs, every byte compared
    assert got[:103] == list(range(103))
    assert len(os.listdir("/proc/self/fd")) < 200


@pytest.mark.parametrize("budget_mb, dont_use_mmap", [(None, False), ("0", False), (None, True)])
def test_mapped_and_plain_reads_agree(dataset, monkeypatch, budget_mb, dont_use_mmap):
    """Persistent mappings (default, file_loader's dont_use_mmap=False), no budget for them, and dont_use_mmap=True: the
    same bytes in the same order over several epochs, also with descriptors being evicted under the mappings."""
    root, files = dataset
    if budget_mb is not None:
        monkeypatch.setenv("DALI_AMD_READER_MMAP_MB", budget_mb)
    monkeypatch.setenv("DALI_AMD_READER_FD_CAP", "5")
    pipe = _pipe(root, 16, prefetch_queue_depth=3, dont_use_mmap=dont_use_mmap)
    got = _take(pipe, 26, files)                          # four epochs, every byte compared
    assert got[:103] == list(range(103))


def test_mapping_budget_falls_back_to_plain_reads(dataset, monkeypatch):
    """0.1 MB of mappings for a larger data set: the files behind the budget are read with pread, every byte still right."""
    root, files = dataset
    total = sum(os.path.getsize(f) for f, _ in files) if isinstance(files[0], tuple) else sum(os.path.getsize(f) for f in files)
    assert total > 200000
    import gc
    gc.collect()                                              # (readers of earlier tests take their mappings with them)
    before = len(_smaps_of(root))
    monkeypatch.setenv("DALI_AMD_READER_MMAP_MB", "0.1")
    pipe = _pipe(root, 16, prefetch_queue_depth=2)
    got = _take(pipe, 20, files)
    assert got[:103] == list(range(103))
    maps = _smaps_of(root)
    mapped = sum(int(l.split()[1]) for l in maps) * 1024      # (page granular: up to 4 KB more than a file's size)
    if before == 0:
        assert 0 < len(maps) < 103 and mapped <= 104858 + 4096 * len(maps)


def _smaps_of(root):
    out, keep = [], False
    for line in open("/proc/self/smaps"):
        head = line.split()
        if head and not head[0].endswith(":"):        # "start-end perms offset dev inode [path]": a new mapping
            keep = len(head) >= 6 and head[-1].startswith(str(root))
        elif keep and line.startswith("Size:"):
            out.append(line)
    return out


def test_a_vanished_file_is_reported_with_its_name(dataset, tmp_path):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    good = dataset[1][0][0]
    gone = str(tmp_path / "gone.jpg")
    open(gone, "wb").write(b"1234")
    pipe = Pipeline(batch_size=2, num_threads=2, device_id=None, prefetch_queue_depth=1)
    with pipe:
        data, label = fn.readers.file(files=[good, gone], name="Reader")
        pipe.set_outputs(data, label)
    pipe.build()
    pipe.run()
    os.remove(gone)
    # the descriptor of the removed file is still open (long-lived descriptors): reading keeps working, as it would with
    # the reference's mmap-ed files; a file that cannot be OPENED is an error that names it
    pipe.run()
    pipe2 = Pipeline(batch_size=2, num_threads=2, device_id=None, prefetch_queue_depth=1)
    with pipe2:
        data, label = fn.readers.file(files=[good, gone], name="Reader")
        pipe2.set_outputs(data, label)
    pipe2.build()
    with pytest.raises(RuntimeError, match="gone.jpg"):
        pipe2.run()


def test_stream_ordered_handover_calls_are_harmless_without_a_device(dataset):
    """Pipeline.share_outputs(cuda_stream=...) / release_outputs(cuda_stream=...) on a CPU-only pipeline: there is nothing
    to order, the outputs are the same as through run()."""
    root, files = dataset
    a, b = _pipe(root, 8, prefetch_queue_depth=2), _pipe(root, 8, prefetch_queue_depth=2)
    for _ in range(5):
        want = a.run()
        b._prefetch()
        b.release_outputs(cuda_stream=0)
        got = b.share_outputs(cuda_stream=0)
        for i in range(8):
            assert got[0].at(i).tobytes() == want[0].at(i).tobytes()
            assert int(np.asarray(got[1].at(i)).reshape(-1)[0]) == int(np.asarray(want[1].at(i)).reshape(-1)[0])
    b.release_outputs(cuda_stream=0)
    b._backend.wait_enqueued()
