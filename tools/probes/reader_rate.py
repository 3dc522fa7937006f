"""GPU box: how fast does readers.file alone hand out 256-file batches (page cache -> page-locked blocks)?"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

enc = bench.make_dataset(0, 1024, workers=bench.effective_cpu_count())
import torch  # noqa: E402,F401
from dali_amd import fn  # noqa: E402
from dali_amd.pipeline import Pipeline  # noqa: E402

root = tempfile.mkdtemp(prefix="reader_rate_")
bench.write_dataset(root, enc)
for rd in (2, 4):
    for env in ({}, {"DALI_AMD_READER_THREADS": "12"}, {"DALI_AMD_READER_THREADS": "4"}):
        os.environ.update(env)
        pipe = Pipeline(batch_size=256, num_threads=12, device_id=0, prefetch_queue_depth=5, set_affinity=True)
        with pipe:
            j, l = fn.readers.file(file_root=root, prefetch_queue_depth=rd)
            pipe.set_outputs(j, l)
        pipe.build()
        for _ in range(30):
            pipe.run()
        t = time.perf_counter()
        for _ in range(400):
            pipe.run()
        el = time.perf_counter() - t
        print(f"reader depth {rd} {env}: {400 * 256 / el:.0f} files/s = {1e3 * el / 400:.3f} ms per batch", flush=True)
        for k in env:
            os.environ.pop(k)
        del pipe
