import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """Host library + oracle are plain g++/gcc builds (seconds); the HIP library is built by
    __graft_entry__.build() and must already exist for the gpu tests."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "dali_amd", "host")])
    if os.environ.get("DALI_AMD_HIPEMU"):
        # developer switch: run the selected tests (the gpu-marked ones included) on the CPU model of the kernels
        # (tools/hipemu); DALI_AMD_HIPEMU=address for the AddressSanitizer build (needs LD_PRELOAD of the runtime)
        from tests import hipemu_env
        san = os.environ["DALI_AMD_HIPEMU"]
        hipemu_env.activate("" if san in ("1", "on") else san)
    yield
