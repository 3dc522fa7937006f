"""Out-of-tree operators: a library compiled against dali_amd/host/framework.h registers its schema + factory when
dali_amd.plugin_manager.load_library dlopens it, and fn / ops grow the new operator (reference:
dali/plugin/plugin_manager.cc:26-41, dali/python/nvidia/dali/plugin_manager.py:19-36)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def plugin(tmp_path_factory):
    d = tmp_path_factory.mktemp("plugin")
    out = str(d / "sub" / "libdali_customops.so")
    os.makedirs(os.path.dirname(out))
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "dali_amd", "host"), os.path.join(ROOT, "tests", "plugin", "custom_ops.cpp"),
                           "-L", os.path.join(ROOT, "dali_amd", "lib"), "-ldali_amd_host",
                           "-Wl,-rpath," + os.path.join(ROOT, "dali_amd", "lib"), "-o", out])
    return str(d), out


def test_load_library_registers_the_operator(plugin):
    from dali_amd import fn, ops, plugin_manager
    from dali_amd.pipeline import Pipeline
    directory, lib = plugin
    assert not hasattr(fn, "custom")
    with pytest.raises(RuntimeError, match="Failed to load library"):
        plugin_manager.load_library(os.path.join(directory, "does_not_exist.so"))
    plugin_manager.load_directory(directory)          # finds sub/libdali_customops.so
    assert "value" in fn.custom.add_constant.__doc__ and ops.custom.AddConstant.schema_name == "custom__AddConstant"
    pipe = Pipeline(batch_size=4, num_threads=2, device_id=0, prefetch_queue_depth=1)
    with pipe:
        x = fn.external_source(name="x")
        pipe.set_outputs(fn.custom.add_constant(x, value=100))
    pipe.build()
    data = [np.arange(i * 50, i * 50 + 12, dtype=np.uint8).reshape(3, 4) for i in range(4)]
    pipe.feed_input("x", data)
    (out,) = pipe.run()
    for i in range(4):
        assert np.array_equal(out.at(i), np.clip(data[i].astype(int) + 100, 0, 255).astype(np.uint8))
    # loading the same library again re-runs nothing (dlopen returns the loaded handle): no duplicate registration
    plugin_manager.load_library(lib)
