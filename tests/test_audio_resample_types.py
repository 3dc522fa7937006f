"""Sample-type handling of fn.audio_resample: the reference's own known answers
(dali/test/python/operator_1/test_audio_resample.py:73-161, `test_dynamic_ranges` and `test_type_conversion`) restated:
constant signals resampled with scale=1 / quality=0 keep their value, so the result is the type conversion alone -
every pair of int8 / uint8 / int16 / uint16 / int32 / uint32 / float, extremes and mid-range values, with the
tolerances of the reference's test.  The CPU backend runs everywhere; the device kernels carry the gpu marker."""
import numpy as np
import pytest

_NP = {"FLOAT": np.float32, "UINT8": np.uint8, "INT8": np.int8, "UINT16": np.uint16, "INT16": np.int16, "INT32": np.int32,
       "UINT32": np.uint32}

DYNAMIC_RANGES = [
    ("FLOAT", [-1.0e30, -1 - 1.0e-6, -1, -0.5, -1.0e-30, 0, 1.0e-30, 0.5, 1, 1 + 1.0e-6, 1e30], 0),
    ("UINT8", [0, 1, 128, 254, 255], 0),
    ("INT8", [-128, -127, -1, 0, 1, 127], 0),
    ("UINT16", [0, 1, 32767, 32768, 65534, 65535], 0),
    ("INT16", [-32768, -32767, -100, -1, 0, 1, 100, 32767], 0),
    ("UINT32", [0, 1, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFE, 0xFFFFFFFF], 128),
    ("INT32", [-0x80000000, -0x7FFFFFFF, -100, -1, 0, 1, 0x7FFFFFFF], 128),
]
TYPE_RANGES = [("FLOAT", [-1, 1]), ("UINT8", [0, 255]), ("INT8", [-127, 127]), ("UINT16", [0, 65535]), ("INT16", [-32767, 32767]),
               ("INT32", [-0x7FFFFFFF, 0x7FFFFFFF]), ("UINT32", [0, 0xFFFFFFFF])]


def _conversion_cases():
    cases = []
    for src, (i_lo, i_hi) in TYPE_RANGES:
        in_values = [i_lo, 0, i_hi] if i_lo == -i_hi else [i_lo, (i_lo + i_hi) // 2, (i_lo + i_hi + 1) // 2, i_hi]
        for dst, (o_lo, o_hi) in TYPE_RANGES:
            if len(in_values) == 3:
                out_values = [o_lo, (o_hi + o_lo + 1) / 2, o_hi] if o_lo != -o_hi else [o_lo, 0, o_hi]
            else:
                out_values = [o_lo, o_lo + (o_hi - o_lo) * in_values[1] / (i_hi - i_lo),
                              o_lo + (o_hi - o_lo) * in_values[2] / (i_hi - i_lo), o_hi]
            if dst != "FLOAT":
                out_values = list(map(int, out_values))
            eps = (o_hi - o_lo) / 2**24 + (i_hi - i_lo) / 2**24
            if eps < 1 and (o_lo != -o_hi or (i_hi != i_lo and dst != "FLOAT")):
                eps = 1   # the result will be halfway
            cases.append((src, in_values, dst, out_values, eps))
    return cases


def _run(device, src, in_values, dst, out_values, eps):
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    in_data = [np.full((100 + 10 * i,), x, _NP[src]) for i, x in enumerate(in_values)]
    pipe = Pipeline(batch_size=len(in_values), num_threads=2, device_id=0 if device == "gpu" else None, prefetch_queue_depth=1)
    with pipe:
        x = fn.external_source(name="x")
        pipe.set_outputs(fn.audio_resample(x.gpu() if device == "gpu" else x, dtype=getattr(types, dst), scale=1, quality=0))
    for _ in range(2):
        pipe.feed_input("x", in_data)
        (out,) = pipe.run()
        for i, want in enumerate(out_values):
            got = out.at(i) if device == "cpu" else out[i].as_cpu()
            got = np.asarray(got)
            ref = np.full_like(in_data[i], want, _NP[dst])
            assert got.dtype == _NP[dst] and got.shape == ref.shape
            assert np.allclose(got.astype(np.float64), ref.astype(np.float64), 1e-6, eps), (src, dst, in_values[i], got[:3], want)


@pytest.mark.parametrize("case", DYNAMIC_RANGES, ids=lambda c: c[0])
def test_dynamic_ranges_cpu(case):
    t, values, eps = case
    _run("cpu", t, values, t, values, eps)


@pytest.mark.parametrize("case", _conversion_cases(), ids=lambda c: f"{c[0]}-{c[2]}")
def test_type_conversion_cpu(case):
    _run("cpu", *case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", DYNAMIC_RANGES, ids=lambda c: c[0])
def test_dynamic_ranges_gpu(case):
    t, values, eps = case
    _run("gpu", t, values, t, values, eps)


@pytest.mark.gpu
@pytest.mark.parametrize("case", _conversion_cases(), ids=lambda c: f"{c[0]}-{c[2]}")
def test_type_conversion_gpu(case):
    _run("gpu", *case)
