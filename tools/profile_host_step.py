"""Development aid: cProfile of bench.HotPath.step (the Python launch path of the resident bench)."""
import cProfile
import pstats
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

enc = bench.make_dataset(0, 256, workers=8)
dev = torch.device("cuda", 0)
hp = bench.HotPath(enc, dev, torch.cuda.Stream(device=dev))
for _ in range(5):
    hp.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    hp.step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
