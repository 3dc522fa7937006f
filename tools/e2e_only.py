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

    def per_thread():
        out = {}
        for tid in os.listdir("/proc/self/task"):
            try:
                comm = open(f"/proc/self/task/{tid}/comm").read().strip()
                ns = int(open(f"/proc/self/task/{tid}/schedstat").read().split()[0])
                st = open(f"/proc/self/task/{tid}/stat").read().rsplit(")", 1)[1].split()
                wchan = open(f"/proc/self/task/{tid}/wchan").read().strip()
                try:
                    sysc = open(f"/proc/self/task/{tid}/syscall").read().split()[0]
                except OSError:
                    sysc = "?"
                out[int(tid)] = (comm, ns * 1e-9, st[0], wchan, sysc, int(st[11]) + int(st[12]))
            except (OSError, ValueError, IndexError):
                pass
        return out
    before = per_thread()
    res = bench.e2e_pipeline(root, kw["batch"], 0, iters=kw["iters"], threads=kw["threads"])
    after = per_thread()
    if os.environ.get("E2E_THREADS"):
        rows = sorted(((after[t][1] - before.get(t, after[t])[1], t) for t in after), reverse=True)[:8]
        for dt, t in rows:
            c = after[t]
            print("      tid %d (%s, %s main) cpu %.3f s = %.3f ms/batch  state %s wchan %s syscall %s" % (
                t, c[0], "is" if t == os.getpid() else "not", dt, 1e3 * dt / kw["iters"], c[2], c[3], c[4]), flush=True)
        print("      threads of the process, in creation order:", [(t, after[t][0]) for t in sorted(after)], flush=True)
    os.sched_setaffinity(0, set(allowed))
    buf = C.create_string_buffer(1 << 16)
    lib.daliamdKernelTimingReport(buf, len(buf))
    lib.daliamdKernelTimingEnable(0)
    kern = {ln.split("\t")[0].replace("Kernel", ""): round(float(ln.split("\t")[2]), 3) for ln in buf.value.decode().splitlines() if ln}
    print("%-60s %8.0f img/s  cpu %.2f ms/batch %s gather %s\n      %s" % (variant or "(default)", res["value"], res["cpu_ms_per_batch"],
                                                                     res["cpu_ms_per_batch_by_thread_group"], "gather_encoded" in res["kernels"], kern), flush=True)
    os.environ.clear()
    os.environ.update(saved)
