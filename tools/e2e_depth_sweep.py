"""GPU box: the end-to-end leg of bench.py (files in the page cache -> fp16 batch) over prefetch_queue_depth x reader depth.
    python tools/e2e_depth_sweep.py [depths...]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

enc = bench.make_dataset(0, 1024, workers=bench.effective_cpu_count())
import torch  # noqa: E402,F401

root = tempfile.mkdtemp(prefix="e2e_sweep_")
bench.write_dataset(root, enc)
depths = [int(a) for a in sys.argv[1:]] or [5, 7, 9, 12]
for rd in (2, 4):
    for d in depths:
        vals = []
        for _ in range(2):
            r = bench.e2e_pipeline(root, 256, 0, iters=400, depth=d, reader_depth=rd)
            vals.append(round(r["value"]))
        print(f"prefetch_queue_depth {d:2d} reader depth {rd}: {vals} img/s  device stage {r['device_stage_ms_per_batch']:.3f} ms host stage "
              f"{r['host_stage_ms_per_batch']:.3f} slot wait {r['slot_wait_ms_per_batch']:.3f}", flush=True)
