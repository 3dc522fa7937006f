"""Development aid: where DALIGenericIterator.__next__ spends its time on the resident headline pipeline (python
tools/iterator_trace.py): pipe.run(), the view of the output, the destination tensor, the copy, the wait."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from dali_amd import _backend  # noqa: E402

enc = bench.make_dataset(0, 1024, workers=8)
root = tempfile.mkdtemp(prefix="dali_amd_it_")
bench.write_dataset(root, enc)
pipe = bench.resident_pipeline(root, 256, 0, 5, 12, cache_mb=256)
while _backend.encoded_cache_stats(0)["streams"] < 1024:
    pipe.run()
for _ in range(30):
    pipe.run()
dev = torch.device("cuda", 0)
for name, side in (("default priority", torch.cuda.Stream(device=dev)), ("high priority", torch.cuda.Stream(device=dev, priority=-1)),
                   ("no copy", None)):
    acc = [0.0] * 5
    n = 100
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for _ in range(n):
        t0 = time.perf_counter()
        data, lab = pipe.run()
        t1 = time.perf_counter()
        if side is not None:
            with torch.cuda.stream(side):
                src = data.as_tensor()
                t2 = time.perf_counter()
                dst = torch.empty(src.shape, dtype=src.dtype, device=src.device)
                t3 = time.perf_counter()
                dst.copy_(src, non_blocking=True)
            t4 = time.perf_counter()
            side.synchronize()
            t5 = time.perf_counter()
            for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                acc[i] += d
        else:
            acc[0] += t1 - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t_all
    print(f"{name:18s} {256 * n / el:9.0f} images/s  ms per step: run {1e3 * acc[0] / n:.3f} view {1e3 * acc[1] / n:.3f} "
          f"empty {1e3 * acc[2] / n:.3f} copy {1e3 * acc[3] / n:.3f} wait {1e3 * acc[4] / n:.3f}", flush=True)
