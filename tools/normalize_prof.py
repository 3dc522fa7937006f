"""Profiling workload for the reduction / normalisation kernels (north_star: "wave64 shuffle reductions for per-channel
mean/std ... evidenced by rocprof achieved HBM GB/s"): fn.normalize(axes=(0, 1)) - NormalizeStatsKernel (mean),
NormalizeStatsKernel (variance), NormalizeFinalizeKernel, NormalizeApplyKernel - and the stand-alone
fn.crop_mirror_normalize (CmnKernel) on 256 decoded-image-sized u8 HWC samples resident in HBM.  One JSON line with the
HIP-event time of every launch and its algorithmic bytes; tools/collect_profiles.sh runs it under rocprofv3
(--kernel-trace --stats, then FETCH_SIZE / WRITE_SIZE passes).  python tools/normalize_prof.py [steps]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from dali_amd import fn, types  # noqa: E402
from dali_amd.pipeline import Pipeline  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = 256
rng = np.random.default_rng(1234)
from dali_amd.testing import IMAGENET_LIKE_SIZES  # noqa: E402
shapes = [IMAGENET_LIKE_SIZES[i % 6] for i in range(N)]
dev = torch.device("cuda", 0)
images = [torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device=dev) for (h, w) in shapes]
pixels = sum(h * w for h, w in shapes)

pipe = Pipeline(batch_size=N, num_threads=4, device_id=0, prefetch_queue_depth=1, seed=1)
with pipe:
    x = fn.external_source(name="x", device="gpu", layout="HWC")
    stats = fn.normalize(x, axes=[0, 1])                                        # per-channel mean / stddev of every sample
    cmn = fn.crop_mirror_normalize(x, crop=[224, 224], dtype=types.FLOAT16, output_layout="CHW",
                                   mean=[0.485 * 255, 0.456 * 255, 0.406 * 255], std=[0.229 * 255, 0.224 * 255, 0.225 * 255],
                                   mirror=1)
    pipe.set_outputs(stats, cmn)
pipe.build()
for _ in range(4):
    pipe.feed_input("x", images)
    pipe.run()
torch.cuda.synchronize()
bench.kernel_timing(8 * (steps + 4))
bench.kernel_timing(False)
bench.kernel_timing()
bench.kernel_timing(True)
for _ in range(steps):
    pipe.feed_input("x", images)
    pipe.run()
torch.cuda.synchronize()
bench.kernel_timing(False)
times = bench.kernel_timing()
# algorithmic bytes per launch: a statistics pass reads every element once; the apply pass reads u8, writes f32;
# the stand-alone CMN reads the 224 x 224 crop and writes fp16
algo = {"NormalizeStatsKernel": 3 * pixels, "NormalizeApplyKernel": 3 * pixels * (1 + 4), "NormalizeFinalizeKernel": 0,
        "CmnKernel": N * 224 * 224 * 3 * (1 + 2)}
per = {}
for name, (calls, ms) in times.items():
    if name in algo:
        per[name] = {"launches": calls, "avg_ms": ms, "algorithmic_bytes": algo[name],
                     "achieved_GBps": algo[name] / (ms * 1e-3) / 1e9 if ms else None,
                     "frac_of_8TBps": algo[name] / (ms * 1e-3) / 8e12 if ms else None}
print(json.dumps({"workload": f"fn.normalize(axes=(0, 1)) + fn.crop_mirror_normalize(crop 224, fp16 CHW) on {N} u8 HWC images "
                              f"({pixels / 1e6:.1f} MPix) resident in HBM", "steps": steps, "kernels": pipe.executed_kernels(),
                  "per_kernel": per,
                  "note": "NormalizeStatsKernel runs twice per iteration (mean, then variance around it): avg over both"}))
