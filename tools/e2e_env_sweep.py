"""GPU box: bench.py's end-to-end leg under environment variants.   python tools/e2e_env_sweep.py "A=1 B=2" "A=0" ..."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

enc = bench.make_dataset(0, 1024, workers=bench.effective_cpu_count())
import torch  # noqa: E402,F401

root = tempfile.mkdtemp(prefix="e2e_env_")
bench.write_dataset(root, enc)
for variant in sys.argv[1:] or [""]:
    saved = dict(os.environ)
    for kv in variant.split():
        k, v = kv.split("=")
        os.environ[k] = v
    vals = []
    for _ in range(3):
        r = bench.e2e_pipeline(root, 256, 0, iters=400)
        vals.append(round(r["value"]))
    print(f"{variant or '(default)':50s} {vals} img/s  device stage {r['device_stage_ms_per_batch']:.3f} host stage {r['host_stage_ms_per_batch']:.3f}", flush=True)
    os.environ.clear()
    os.environ.update(saved)
