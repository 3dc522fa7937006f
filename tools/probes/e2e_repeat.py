"""GPU box: the end-to-end leg four times in one process - does a later repetition run slower than the first, and does collecting
the earlier pipeline (or a pause) bring the rate back?"""
import gc
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

enc = bench.make_dataset(0, 1024, workers=bench.effective_cpu_count())
import torch  # noqa: E402,F401

root = tempfile.mkdtemp(prefix="e2e_repeat_")
bench.write_dataset(root, enc)
for mode in ("plain", "gc", "gc+sleep"):
    vals = []
    for _ in range(4):
        r = bench.e2e_pipeline(root, 256, 0, iters=400)
        vals.append(round(r["value"]))
        if "gc" in mode:
            gc.collect()
            torch.cuda.empty_cache()
        if "sleep" in mode:
            time.sleep(2.0)
    print(mode, vals, "threads alive:", len(os.listdir("/proc/self/task")), flush=True)
