"""GPU box: the headline graph on one of the realistic data set variants (dali_amd.testing.DATASET_VARIANTS), with the
kernel library's per-launch timing: which kernel pays for 12-megapixel outliers / per-file tables / host-decoded streams.
    python tools/variant_kernels.py VARIANT [--inflight D] [--steps K] [--images N]"""
import argparse
import json
import os
import shutil
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("variant")
ap.add_argument("--inflight", type=int, default=5)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--images", type=int, default=1024)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--cache-type", default="encoded")
args = ap.parse_args()
enc = bench.make_dataset(0, args.images, workers=bench.effective_cpu_count(), variant=args.variant)
import torch  # noqa: E402
torch.cuda.set_device(0)
root = tempfile.mkdtemp(prefix="dali_amd_variant_")
try:
    bench.write_dataset(root, enc)
    bench.kernel_timing(16 * (args.steps + 64))
    bench.kernel_timing(False)
    orig = bench.resident_variant_leg

    res = bench.resident_variant_leg(args, root, len(enc), sum(map(len, enc)), 0, args.steps, cache_type=args.cache_type,
                                     time_kernels=True)
    print(json.dumps(res, indent=1))
finally:
    shutil.rmtree(root, ignore_errors=True)
