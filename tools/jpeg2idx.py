#!/usr/bin/env python
"""Offline index of a JPEG data set for the GPU entropy decoder (round 6; the reference indexes ITS containers offline:
tools/tfrecord2idx, tools/wds2idx.py, tools/rec2idx.py).

    python tools/jpeg2idx.py FILE_ROOT INDEX_ROOT [--workers N] [--filters "*.jpg" "*.jpeg" "*.JPEG"]

For every baseline JPEG below FILE_ROOT writes INDEX_ROOT/<relative name>.didx: the file's headers + the index entry of its
entropy-coded segment (un-stuffed stream + 12 bytes of decoder state per 256-byte slice), built with the HOST restatement of
the decoder's position pass (daliamdJpegIndexedBuild -> daliamdJpegHuffmanIndexBuildHost).  No GPU needed.  Then

    fn.readers.file(file_root=FILE_ROOT, index_path=INDEX_ROOT)  ->  fn.decoders.image(device="mixed")

reads the containers in place of the files and decodes them from the index - the decoder's un-stuffing, relaxation, hand-over
and DC passes do not run, in the first epoch and in a cold process too.  Files the GPU decoder does not take (progressive,
restart intervals, CMYK, PNG ...) get no container and are read as they are.  A container is 4-6 % larger than its file."""
import argparse
import ctypes as C
import fnmatch
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_one(job):
    src, dst = job
    import numpy as np
    from dali_amd import _capi as capi
    host = capi.host()
    data = np.fromfile(src, np.uint8)
    n = C.c_size_t(0)
    if host.daliamdJpegIndexedBuild(data.ctypes.data_as(C.c_void_p), C.c_size_t(data.size), None, C.c_size_t(0), C.byref(n)) != 0:
        return src, 0, (host.daliamdHostGetLastErrorMessage() or b"").decode()
    out = np.empty(n.value, np.uint8)
    if host.daliamdJpegIndexedBuild(data.ctypes.data_as(C.c_void_p), C.c_size_t(data.size), out.ctypes.data_as(C.c_void_p),
                                    C.c_size_t(out.size), C.byref(n)) != 0:
        return src, 0, (host.daliamdHostGetLastErrorMessage() or b"").decode()
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    tmp = dst + ".tmp"
    out[:n.value].tofile(tmp)
    os.replace(tmp, dst)
    return src, int(n.value), ""


def index_tree(file_root, index_root, workers=0, filters=("*.jpg", "*.jpeg", "*.JPG", "*.JPEG"), quiet=False, threads=False):
    """threads: a thread pool instead of forked workers (the library calls release the GIL) - for a process that has already
    initialised the GPU runtime."""
    jobs = []
    for dirpath, _, files in os.walk(file_root):
        for f in sorted(files):
            if any(fnmatch.fnmatch(f, p) for p in filters):
                src = os.path.join(dirpath, f)
                jobs.append((src, os.path.join(index_root, os.path.relpath(src, file_root) + ".didx")))
    if threads and workers and workers > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as pool:
            results = list(pool.map(build_one, jobs))
    elif workers and workers > 1 and len(jobs) >= 2 * workers:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(workers) as pool:
            results = pool.map(build_one, jobs, chunksize=max(1, len(jobs) // (8 * workers)))
    else:
        results = [build_one(j) for j in jobs]
    made = [r for r in results if r[1]]
    skipped = [r for r in results if not r[1]]
    if not quiet:
        print(f"jpeg2idx: {len(made)} containers, {sum(r[1] for r in made) / 1e6:.1f} MB; {len(skipped)} files left as they are")
        for src, _, why in skipped[:10]:
            print(f"  {src}: {why}")
    return len(made), len(skipped)


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("file_root")
    ap.add_argument("index_root")
    ap.add_argument("--workers", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--filters", nargs="*", default=["*.jpg", "*.jpeg", "*.JPG", "*.JPEG"])
    a = ap.parse_args()
    index_tree(a.file_root, a.index_root, a.workers, tuple(a.filters))
