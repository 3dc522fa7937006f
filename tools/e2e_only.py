"""GPU box: bench.py's end-to-end leg alone, under a list of environment variants, with the kernel library's per-kernel
timing switched on (durations inside the schedule).   python tools/e2e_only.py "A=1 B=2" "A=0" ..."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

enc = bench.make_dataset(0, 2048, workers=8)
import ctypes as C  # noqa: E402
import torch  # noqa: E402,F401
from dali_amd import _capi as capi  # noqa: E402

root = tempfile.mkdtemp(prefix="e2e_only_")
bench.write_dataset(root, enc)
lib = capi.kernels()
lib.daliamdKernelTimingReport.argtypes = [C.c_char_p, C.c_int]
for variant in sys.argv[1:] or [""]:
    saved = dict(os.environ)
    for kv in variant.split():
        k, v = kv.split("=")
        os.environ[k] = v
    kw = dict(batch=256, iters=200, threads=None)
    allowed = sorted(os.sched_getaffinity(0))
    if os.environ.get("E2E_WORLD8", "0") != "0":   # bench.py's e2e_pipeline_local_world8: 2 of the usable CPUs, batch 512, thread pools of 2
        os.sched_setaffinity(0, set(allowed[:2]))
        kw = dict(batch=512, iters=100, threads=2)
    bench.e2e_pipeline(root, kw["batch"], 0, iters=60, threads=kw["threads"])                       # warm: mappings made and registered
    lib.daliamdKernelTimingEnable(4096)
    res = bench.e2e_pipeline(root, kw["batch"], 0, iters=kw["iters"], threads=kw["threads"])
    os.sched_setaffinity(0, set(allowed))
    buf = C.create_string_buffer(1 << 16)
    lib.daliamdKernelTimingReport(buf, len(buf))
    lib.daliamdKernelTimingEnable(0)
    kern = {ln.split("\t")[0].replace("Kernel", ""): round(float(ln.split("\t")[2]), 3) for ln in buf.value.decode().splitlines() if ln}
    print("%-60s %8.0f img/s  cpu %.2f ms/batch %s gather %s\n      %s" % (variant or "(default)", res["value"], res["cpu_ms_per_batch"],
                                                                     res["cpu_ms_per_batch_by_thread_group"], "gather_encoded" in res["kernels"], kern), flush=True)
    os.environ.clear()
    os.environ.update(saved)
