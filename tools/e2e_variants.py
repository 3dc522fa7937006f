"""Development aid: the end-to-end leg of bench.py (files in the page cache -> fp16 batch on the device) under different
host settings - executor threads, reader threads, reader prefetch depth, CPU affinity - with the CPU account per thread
group.  One line per variant.  python tools/e2e_variants.py [quick]"""
import json
import os
import shutil
import sys
import tempfile

sys.path.insert(0, ".")
import bench  # noqa: E402

enc = bench.make_dataset(0, 1024, workers=8)
root = tempfile.mkdtemp(prefix="dali_amd_e2e_")
bench.write_dataset(root, enc)
variants = [dict(), dict(reader_depth=2), dict(reader_depth=3), dict(reader_threads=6), dict(reader_threads=8, reader_depth=2),
            dict(reader_threads=16, reader_depth=2), dict(threads=8, reader_depth=2), dict(threads=8, reader_threads=8, reader_depth=2),
            dict(threads=6, reader_threads=10, reader_depth=2), dict(set_affinity=True, reader_depth=2),
            dict(depth=5, reader_depth=2), dict(depth=6, reader_depth=3), dict(cache_mb=1024, reader_depth=2), dict(roi_decode=True, reader_depth=2)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    variants = variants[:3]
for v in variants:
    kw = dict(v)
    rt = kw.pop("reader_threads", None)
    if rt:
        os.environ["DALI_AMD_READER_THREADS"] = str(rt)
    else:
        os.environ.pop("DALI_AMD_READER_THREADS", None)
    r = bench.e2e_pipeline(root, 256, 0, iters=300, **kw)
    print(json.dumps({"variant": v, "images_per_s": round(r["value"]), "ms_per_batch": round(r["ms_per_batch"], 4),
                      "reader_wait_ms": r["host_ms_per_operator"].get("Reader"),
                      "decoder_host_ms": [x for k, x in r["host_ms_per_operator"].items() if "decoders" in k],
                      "cpu_ms": r["cpu_ms_per_batch_by_thread_group"], "cpus_busy": r["cpus_busy"]}), flush=True)
shutil.rmtree(root, ignore_errors=True)
