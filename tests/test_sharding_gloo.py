"""Multi-process path on CPU (gloo, world_size 2 and 8): every rank builds the same reader pipeline with
shard_id=rank / num_shards=world, exactly as bench.py and a DDP training script do.  The ranks exchange what
they read (test-only all_gather -- the data path itself has no collective) and check that the shards are
disjoint, cover the dataset, and rotate between epochs."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, sys.argv[1])
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    root = sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pipe = Pipeline(batch_size=4, num_threads=2, device_id=0, seed=1)
    with pipe:
        d, l = fn.readers.file(file_root=root, shard_id=rank, num_shards=world, pad_last_batch=True, name="Reader")
        pipe.set_outputs(d, l)
    meta = pipe.reader_meta("Reader")
    per_epoch = meta["epoch_size_padded"] // world
    epochs = []
    for e in range(2):
        idx = []
        for _ in range(per_epoch // 4):
            d, l = pipe.run()
            idx += [int(np.frombuffer(d.at(i).tobytes(), np.int32)[0]) for i in range(4)]
        epochs.append(idx)
    gathered = [None] * world
    dist.all_gather_object(gathered, epochs)
    if rank == 0:
        n = meta["epoch_size"]
        for e in range(2):
            allidx = sorted(set(sum((g[e] for g in gathered), [])))
            assert allidx == list(range(n)), (e, allidx)
            for r in range(world):
                shard = (r + e) % world
                lo, hi = n * shard // world, n * (shard + 1) // world
                assert sorted(set(gathered[r][e])) == list(range(lo, hi)), (e, r, gathered[r][e])
        print("SHARDING_OK", n, per_epoch)
    dist.barrier()
    dist.destroy_process_group()
''')


import pytest


@pytest.mark.parametrize("world,classes,expected", [(2, (("a", 7), ("b", 10), ("c", 6)), "SHARDING_OK 23 12"),
                                                    # the driver's largest launch: 8 ranks, shards of 11 and 12 samples (the short ones padded)
                                                    (8, (("a", 40), ("b", 30), ("c", 22)), "SHARDING_OK 92 12")])
def test_sharded_readers(tmp_path, world, classes, expected):
    root = tmp_path / "ds"
    k = 0
    for c, n in classes:
        os.makedirs(root / c)
        for i in range(n):
            (root / c / f"f{i:02d}.jpg").write_bytes(np.int32(k).tobytes())
            k += 1
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29533 + world), str(script), ROOT, str(root)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    assert expected in res.stdout, res.stdout[-500:]
