"""The device code on a CPU model (tools/hipemu): the kernel sources of dali_amd/csrc - unchanged - compiled as C++
against an execution model of the HIP constructs they use (a fiber per lane, LDS per workgroup, wave64 shuffles /
ballots / readfirstlane with divergence, MFMA 16x16x4, atomics, synchronous streams), linked under the product's host
library, and the gpu-marked parity tests run against that in a container without a GPU.

What it proves: the kernels' arithmetic, indexing, LDS choreography and barrier placement give the oracle's results
(bit-exact where the gpu tests ask for that), and - in the AddressSanitizer build, DALI_AMD_HIPEMU=address - that no
kernel touches a byte outside its buffers.  What it cannot: anything about time, and hardware behaviour the model does
not have (memory ordering between waves without a barrier, LDS bank conflicts).  The gpu tests on the MI355X remain the
parity gate; this is the same gate one stage earlier.  TEST INFRASTRUCTURE: nothing under dali_amd/ knows the model
exists, and the subprocesses below are the only place that loads it.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# tests that assert properties of the real device runtime (torch's device type of a DLPack capsule, .is_cuda) or start
# their own python without the model
NOT_ON_THE_MODEL = [
    "tests/test_gpu_augment.py::test_blur_fma_variant_stays_within_the_reference_tolerance",
    "tests/test_gpu_pipeline.py::test_iterator_two_shards",
    "tests/test_gpu_pipeline.py::test_gpu_tensor_dlpack_zero_copy_and_device_feed",
]

GROUPS = {
    "jpeg": ["tests/test_gpu_jpeg.py"],
    "resample_cmn_normalize": ["tests/test_gpu_resample.py", "tests/test_gpu_cmn.py", "tests/test_gpu_normalize.py"],
    "augment_audio": ["tests/test_gpu_augment.py", "tests/test_gpu_audio.py", "tests/test_output_types.py",
                      "tests/test_gpu_formats.py"],
    "pipelines": ["tests/test_gpu_config1.py", "tests/test_gpu_roi_resize.py", "tests/test_gpu_pipeline.py",
                  "tests/test_gpu_decoder_cache.py", "tests/test_gpu_encoded_cache.py", "tests/test_gpu_roi_fusion.py",
                  "tests/test_gpu_jpeg_indexed_files.py", "tests/test_gpu_reader_zero_copy.py"],   # (round 6)
}
if os.environ.get("DALI_AMD_HIPEMU_FULL"):
    GROUPS["headline"] = ["tests/test_gpu_headline.py"]   # b256 x 3 epochs + b512 through the bench's graph: 85 s


def _build():
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "tools", "hipemu")])


def test_execution_model_selftest():
    """Scans over shuffles, ballots in loops that lanes leave at different times, collectives in both arms of a branch,
    butterfly reductions + atomics over several OS threads, dynamic LDS, the MFMA lane layout, 3-D launches."""
    _build()
    out = subprocess.run([os.path.join(ROOT, "tools", "hipemu", "_build", "selftest")], capture_output=True, text=True)
    assert out.returncode == 0 and "selftest OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("group", sorted(GROUPS))
def test_gpu_parity_tests_pass_on_the_cpu_model(group):
    _build()
    env = dict(os.environ, DALI_AMD_HIPEMU="1")
    env.pop("LD_PRELOAD", None)
    cmd = [sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-p", "no:xdist"] + GROUPS[group]
    for t in NOT_ON_THE_MODEL:
        if t.split("::")[0] in GROUPS[group]:
            cmd += ["--deselect", t]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = "\n".join(out.stdout.splitlines()[-40:]) + out.stderr[-2000:]
    assert out.returncode == 0, tail
    assert " passed" in out.stdout and "failed" not in out.stdout.splitlines()[-1], tail


# Initcheck rides along: a read of LDS bytes that no lane of the running workgroup has written (what the device hands out
# there is whatever the previous workgroup left) is reported AND served with all-ones bytes (HIPEMU_LDS_POISON, NaN for
# floats) - the kernels that do such reads on purpose (register-blocked passes that overrun a tile's valid part, padded
# MFMA tiles, the aligned dwords around a tap) must still give the oracle's bytes, which is what the tests of the group
# assert.
# Racecheck (make RACE=1): the kernel sources carry ThreadSanitizer's access hooks, answered by the model's lane-level
# detector - two lanes touching the same bytes, one of them writing, with no workgroup barrier (different waves) or no
# wave-wide operation (same wave) in between.  Findings that are not defects, by the text of a source line or the name
# of an (inlined) function of one of the two accesses (line numbers move):
BENIGN_RACES = {
    # LdsDwordAt reads the two aligned dwords around a 4-byte tap; of the last tap of the staged window only three
    # bytes are used, the rest of its second dword can belong to the fp32 intermediate behind the window that other
    # waves are already writing.  The funnel shift drops those bytes.
    "LdsDwordAt",
    # the lanes of the blocks of one colour component all store that component's pitch: same value
    "G.comp_pitch[comp] = d.plane_pitch[comp];",
}
RACE_GROUPS = {"race_resample_cmn_normalize": ["tests/test_gpu_resample.py", "tests/test_gpu_cmn.py",
                                               "tests/test_gpu_normalize.py"]}
if os.environ.get("DALI_AMD_HIPEMU_FULL"):
    RACE_GROUPS["race_jpeg"] = ["tests/test_gpu_jpeg.py", "tests/test_gpu_roi_resize.py", "tests/test_gpu_config1.py"]
    RACE_GROUPS["race_augment_audio"] = ["tests/test_gpu_augment.py", "tests/test_gpu_audio.py"]


def test_racecheck_selftest():
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "tools", "hipemu"), "RACE=1"])
    out = subprocess.run([os.path.join(ROOT, "tools", "hipemu", "_build_race", "racetest")], capture_output=True, text=True)
    assert out.returncode == 0 and "racetest OK" in out.stdout, out.stdout + out.stderr[-2000:]
    for kind in ("different waves, no barrier", "lanes of one wave", "write-write", "uninitialised read of static LDS"):
        assert kind in out.stderr, kind


@pytest.mark.parametrize("group", sorted(RACE_GROUPS))
def test_kernels_are_race_free_on_the_cpu_model(group):
    import re
    emu = os.path.join(ROOT, "tools", "hipemu")
    subprocess.check_call(["make", "-s", "-j8", "-C", emu, "RACE=1"])
    env = dict(os.environ, DALI_AMD_HIPEMU="race")
    env.pop("LD_PRELOAD", None)
    cmd = [sys.executable, "-m", "pytest", "-q", "-s", "-m", "gpu", "-p", "no:cacheprovider", "-p", "no:xdist"] + RACE_GROUPS[group]
    for t in NOT_ON_THE_MODEL:
        if t.split("::")[0] in RACE_GROUPS[group]:
            cmd += ["--deselect", t]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=3000)
    assert out.returncode == 0, "\n".join(out.stdout.splitlines()[-40:]) + out.stderr[-2000:]
    lib = os.path.join(emu, "_build_race", "lib", "libdali_amd_kernels.so")
    pairs = set(re.findall(r"racecheck: (\S+) race .*? at \+(0x[0-9a-f]+), then lane \d+ at \+(0x[0-9a-f]+)", out.stdout + out.stderr))
    defects = []
    for kind, first, second in sorted(pairs):
        sym = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-symbolizer", "-e", lib, first, second], capture_output=True,
                             text=True).stdout.split("\n\n")
        # file:line:col of every (inlined) frame of the two accesses
        where = [ln for blk in sym for ln in blk.strip().splitlines() if re.search(r":\d+:\d+$", ln)]
        texts = [ln.strip() for blk in sym for ln in blk.strip().splitlines() if not re.search(r":\d+:\d+$", ln)]
        for w in where:
            path, line = w.rsplit(":", 2)[0], int(w.rsplit(":", 2)[1])
            texts.append(open(path).read().splitlines()[line - 1].strip() if line > 0 and os.path.exists(path) else "?")
        if not any(t in BENIGN_RACES for t in texts):
            defects.append((kind, where, texts))
    assert not defects, defects


def test_device_decoder_survives_damaged_streams():
    """tools/fuzz_gpu_decoder.py: mutated entropy-coded segments / tables through the GPU entropy decoder on the
    AddressSanitizer build of the model with zero slack behind the allocations - every batch decodes or raises, no kernel
    leaves its buffers, none loops (on the device: silent corruption of HBM, or a hung GPU)."""
    env = dict(os.environ)
    env.pop("DALI_AMD_HIPEMU", None)
    # FUZZ_CONTAINERS (round 6): indexed JPEG containers with damage anywhere - their index entries come from a FILE and reach
    # the device as they are
    runs = [("250", "5", {}), ("120", "9", {"FUZZ_CONTAINERS": "1"})]
    if os.environ.get("DALI_AMD_HIPEMU_FULL"):
        runs += [("1500", "7", {}), ("150", "3", {"FUZZ_BIG": "1"})]
    for iterations, seed, extra in runs:
        out = subprocess.run([os.path.join(ROOT, "tools", "fuzz_gpu_decoder.sh"), iterations, seed], cwd=ROOT, env=dict(env, **extra),
                             capture_output=True, text=True, timeout=3000)
        assert out.returncode == 0 and "fuzz_gpu_decoder OK" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


def test_smoke_and_bench_scripts_run_on_the_cpu_model():
    """Dry runs of the two scripts the driver executes on the MI355X, with the kernels on the model: __graft_entry__.smoke()
    (its own comparison against the oracle included) and - DALI_AMD_HIPEMU_FULL, 2 minutes - the default bench.py line with
    every leg at a small batch: the scripts' host paths still run end to end and print what the driver parses."""
    import json
    env = dict(os.environ)
    env.pop("DALI_AMD_HIPEMU", None)
    runner = [sys.executable, os.path.join(ROOT, "tools", "hipemu", "run_on_model.py")]
    out = subprocess.run(runner + ["-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0 and "smoke OK" in out.stdout, out.stdout[-1000:] + out.stderr[-3000:]
    if not os.environ.get("DALI_AMD_HIPEMU_FULL"):
        return
    # (--side-legs --full-line: every leg, the whole details object on stdout; --no-variants: no 12-megapixel decodes on the model)
    out = subprocess.run(runner + ["bench.py", "--steps", "2", "--warmup", "1", "--batch", "8", "--batches", "2", "--e2e-batch", "8",
                                   "--side-legs", "--full-line", "--no-variants"],
                         cwd=ROOT, env=dict(env, BENCH_DETAILS=os.devnull), capture_output=True, text=True, timeout=3000)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "heavy_aug", "audio", "e2e_pipeline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["roofline"]["bound"] == "hbm"
    for leg in ("heavy_aug", "audio"):
        assert "roofline" in line[leg] and "cpu_baseline" in line[leg], leg


@pytest.mark.parametrize("ranks", [2, 8] if os.environ.get("DALI_AMD_HIPEMU_FULL") else [2])
def test_multi_rank_bench_launch_runs_on_the_cpu_model(ranks, tmp_path):
    """The driver's N > 1 launch of bench.py (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N), every
    rank with the kernels on the model and a gloo group (BENCH_TEST_SINGLE_DEVICE=1, as tests/test_bench_multirank.py does on
    one real device): the sharded pipelines build, the barrier / max-over-ranks timing runs, rank 0 prints ONE line whose
    value counts every rank's batches.  8 ranks - the driver's largest launch - under DALI_AMD_HIPEMU_FULL (70 s)."""
    import json
    details = str(tmp_path / "details.json")
    env = dict(os.environ, BENCH_TEST_SINGLE_DEVICE="1", MASTER_ADDR="127.0.0.1", HIPEMU_THREADS=str(max(1, 8 // ranks)),
               BENCH_DETAILS=details)
    env.pop("DALI_AMD_HIPEMU", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(29580 + ranks), os.path.join(ROOT, "tools", "hipemu", "run_on_model.py"), "bench.py", "--gpus", str(ranks),
           "--steps", "2", "--warmup", "1", "--batch", "8", "--batches", "2", "--e2e-batch", "8"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=3000)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"rank 0 must print exactly one JSON line, got {len(lines)}"
    line = json.loads(lines[0])
    assert line["n_gpus"] == ranks and line["scaling"] == "weak" and line["config"]["global_batch"] == 8 * ranks
    assert f"shard{ranks}" in line["config"]["parallelism"]
    assert abs(line["value"] - ranks * 8 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]
    assert len(lines[0]) < 4000 and out.stdout.strip().splitlines()[-1] == lines[0]      # the compact line, last on stdout
    assert json.load(open(details))["e2e_pipeline_sharded"]["num_shards"] == ranks and line["config"]["e2e_sharded_images_per_s"] > 0


def test_the_product_does_not_know_the_model():
    """No file of the product or of the timed benchmark mentions the model or its libraries."""
    for base in ("dali_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            if "__pycache__" in dirpath or os.sep + "build" in dirpath:
                continue
            for f in files:
                if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="replace").read().lower()
                    assert "hipemu" not in text, os.path.join(dirpath, f)
    for f in ("bench.py", "__graft_entry__.py"):
        assert "hipemu" not in open(os.path.join(ROOT, f)).read().lower(), f
