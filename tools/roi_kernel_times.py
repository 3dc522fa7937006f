"""Per-kernel durations (one batch in flight and five) of the resident headline graph and of its region-of-interest variant
(decoders.image_random_crop -> resize): where the ROI decode saves time and where it does not.
    python tools/roi_kernel_times.py [batches]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    from dali_amd.testing import synth_dataset
    B, nb = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 4
    enc = synth_dataset(0, B * nb, workers=8)
    root = tempfile.mkdtemp(prefix="dali_amd_roi_")
    bench.write_dataset(root, enc)
    out = {}
    for roi in (False, True):
        for depth in (1, 5):
            pipe = bench.resident_pipeline(root, B, 0, depth, 12, cache_mb=max(64, int(2 * sum(len(e) for e in enc) / 2**20)),
                                           roi_decode=roi)
            for _ in range((depth + 3) * nb + 8):
                pipe.run()
            torch.cuda.synchronize()
            bench.kernel_timing(True)
            t0 = time.perf_counter()
            for _ in range(10 * nb):
                pipe.run()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            bench.kernel_timing(False)
            k = {n: round(ms, 4) for n, (calls, ms) in bench.kernel_timing().items()}
            out[f"{'roi' if roi else 'full'}_depth{depth}"] = {"images_per_s": round(10 * nb * B / el), "kernels_ms": k,
                                                                 "sum_ms": round(sum(k.values()), 4)}
            del pipe
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
