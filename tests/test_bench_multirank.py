"""bench.py's N > 1 code path on a ONE-GPU box (gpu-marked): two ranks launched the way the driver launches them
(python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2), both on cuda:0 with a gloo group
(BENCH_TEST_SINGLE_DEVICE=1).  What is checked is the path, not the numbers: every rank builds its sharded pipelines
(readers.file(shard_id=rank, num_shards=2)), the barrier / max-over-ranks timing runs, rank 0 prints ONE JSON line with
n_gpus == 2 whose value counts both ranks' batches, and the sharded end-to-end leg (configs[4]) is present."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_device_produce_one_sharded_line(tmp_path):
    details_path = str(tmp_path / "details.json")
    env = dict(os.environ, BENCH_TEST_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1",
               BENCH_DETAILS=details_path)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--batch", "64", "--batches", "2", "--e2e-batch", "64"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"rank 0 must print exactly one JSON line, got {len(lines)}"
    assert res.stdout.strip().splitlines()[-1] == lines[0] and len(lines[0]) < 4000   # the driver parses the LAST line
    line = json.loads(lines[0])
    details = json.load(open(details_path))
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 128 and "shard2" in line["config"]["parallelism"]
    assert abs(line["value"] - 2 * 64 * 4 / (line["ms_per_step"] * 4e-3)) < 1e-6 * line["value"]
    sharded = details["e2e_pipeline_sharded"]
    assert sharded["num_shards"] == 2 and sharded["value"] > 0 and sharded["elapsed_s_max_over_ranks"] >= sharded["elapsed_s"] - 1e-9
    assert abs(line["config"]["e2e_sharded_images_per_s"] - sharded["value"]) <= 1e-5 * sharded["value"]
    assert details["config"]["pipeline"]["encoded_cache"]["streams"] == 128      # rank 0's shard, resident
    assert "threads per rank" in res.stderr                                    # the thread split is printed
