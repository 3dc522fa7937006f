"""The C-ABI libraries load and export every symbol the headers declare (no compute calls: no GPU here)."""
import ctypes as C
import os
import re

from dali_amd import _capi as capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, macro):
    text = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(macro + r"\s+[\w\s\*]+?\b(daliamd\w+)\s*\(", text)))


def test_kernel_library_exports_all_declared_symbols():
    names = _declared("dali_amd_kernels.h", "DALIAMD_API")
    assert len(names) >= 30
    assert sorted(capi.declared_kernel_symbols()) == names, "python binding list out of sync with the header"
    lib = capi.kernels()
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/dali_amd_kernels.h but not exported"
    assert lib.daliamdVersion() >= 100


def test_host_library_exports_all_declared_symbols():
    names = _declared("dali_amd_host.h", "DALIAMD_HOST_API")
    assert sorted(capi.declared_host_symbols()) == names
    lib = capi.host()
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/dali_amd_host.h but not exported"


def test_pipeline_c_api_exports_all_declared_symbols_and_nothing_undeclared():
    """include/dali_amd_pipeline.h: the flat pipeline API the Python front end binds (dali_amd/_backend.py)."""
    names = _declared("dali_amd_pipeline.h", "DALIAMD_PIPE_API")
    assert len(names) >= 35
    lib = capi.host()
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/dali_amd_pipeline.h but not exported"
    src = open(os.path.join(ROOT, "dali_amd", "host", "c_api.cpp")).read()
    defined = set(re.findall(r"^API\s+[\w\s\*]+?\b(daliamd\w+)\s*\(", src, re.M))
    assert defined == set(names), sorted(defined ^ set(names))


def test_struct_sizes_match_the_c_headers():
    """ctypes mirrors are compiled against: build a tiny C program printing sizeof() of each struct."""
    import subprocess
    import tempfile
    src = r'''
#include <stdio.h>
#include "dali_amd_kernels.h"
#include "dali_amd_host.h"
int main() {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(daliamdJpegIdctDesc), sizeof(daliamdJpegColorDesc),
         sizeof(daliamdResampleArgs), sizeof(daliamdResampleDesc), sizeof(daliamdCmnDesc), sizeof(daliamdJpegInfo),
         sizeof(daliamdPhiloxState), sizeof(daliamdWarpAffineDesc), sizeof(daliamdGaussianBlurDesc),
         sizeof(daliamdPointwiseDesc), sizeof(daliamdJpegHuffDesc), sizeof(daliamdJpegScan),
         sizeof(daliamdJpegRoiPlan), sizeof(daliamdNormalizeDesc), sizeof(daliamdGatherDesc));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "s")]).split()]
    mirrors = [capi.JpegIdctDesc, capi.JpegColorDesc, capi.ResampleArgs, capi.ResampleDesc, capi.CmnDesc,
               capi.JpegInfo, capi.PhiloxState, capi.WarpAffineDesc, capi.GaussianBlurDesc, capi.PointwiseDesc,
               capi.JpegHuffDesc, capi.JpegScan, capi.JpegRoiPlan, capi.NormalizeDesc, capi.GatherDesc]
    assert sizes == [C.sizeof(m) for m in mirrors]


def test_setup_functions_validate_arguments():
    lib = capi.kernels()
    import numpy as np
    args = np.zeros(1, np.dtype(capi.ResampleArgs))
    descs = np.zeros(1, np.dtype(capi.ResampleDesc))
    plan = capi.ResamplePlan()
    rc = lib.daliamdResampleSetup(args.ctypes.data_as(C.c_void_p), 1, descs.ctypes.data_as(C.c_void_p), C.byref(plan))
    assert rc == 1 and b"empty" in lib.daliamdGetLastErrorMessage()
    a = args[0]
    a["in_h"], a["in_w"], a["channels"], a["in_pitch"], a["out_h"], a["out_w"] = 100, 80, 3, 240, 20, 30
    a["min_filter"], a["mag_filter"], a["antialias"] = 1, 1, 1
    rc = lib.daliamdResampleSetup(args.ctypes.data_as(C.c_void_p), 1, descs.ctypes.data_as(C.c_void_p), C.byref(plan))
    assert rc == 0 and plan.num_tiles == descs[0]["tiles_x"] * descs[0]["tiles_y"] > 0 and 0 < plan.lds_bytes <= 60 * 1024
    # the per-sample tables: first tap + coefficients of 30 columns and 20 rows
    assert plan.table_entries == 50
    assert plan.workspace_bytes >= 4 * (30 * (1 + descs[0]["support"][0]) + 20 * (1 + descs[0]["support"][1]))
    # i16 input: the two-launch path - no tiles, an fp32 intermediate, one item per element and pass
    a["in_dtype"], a["out_dtype"], a["in_pitch"] = capi.INT16, capi.INT16, 480
    rc = lib.daliamdResampleSetup(args.ctypes.data_as(C.c_void_p), 1, descs.ctypes.data_as(C.c_void_p), C.byref(plan))
    assert rc == 0 and plan.num_tiles == 0 and descs[0]["generic"] == 1 and plan.generic_items[1] == 20 * 30 * 3
    assert plan.generic_items[0] == descs[0]["tmp_w"] * descs[0]["tmp_h"] * 3 > 0
    a["in_dtype"], a["out_dtype"], a["in_pitch"] = capi.UINT8, capi.UINT8, 240
    a["channels"] = 7
    rc = lib.daliamdResampleSetup(args.ctypes.data_as(C.c_void_p), 1, descs.ctypes.data_as(C.c_void_p), C.byref(plan))
    assert rc == 2  # DALIAMD_ERROR_UNSUPPORTED
