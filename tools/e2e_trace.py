"""Development aid: the end-to-end pipeline of bench.py alone, with DALI_AMD_TRACE=1 (host time per operator)."""
import os
import shutil
import sys
import tempfile

os.environ["DALI_AMD_TRACE"] = "1"
sys.path.insert(0, ".")
import bench  # noqa: E402

enc = bench.make_dataset(0, 1024, workers=8)
root = tempfile.mkdtemp(prefix="dali_amd_e2e_")
bench.write_dataset(root, enc)
for threads in (16, 12, 16, 8):
    r = bench.e2e_pipeline(root, 256, 0, iters=100, threads=threads)
    print("threads", threads, round(r["value"]), r["ms_per_batch"], flush=True)
shutil.rmtree(root, ignore_errors=True)
