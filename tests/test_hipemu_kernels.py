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
                  "tests/test_gpu_decoder_cache.py", "tests/test_gpu_encoded_cache.py"],
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
