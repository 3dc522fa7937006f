"""fn.normalize on the CPU backend (host kernel with the device kernels' arithmetic) against the numpy oracle, the
cases of tests/test_gpu_normalize.py.  Tolerance: rtol = atol = 1e-3 like the reference's own test
(dali/test/python/operator_1/test_normalize.py:209-211); the fp64 accumulation keeps the real error around 1e-6."""
import numpy as np
import pytest

from oracle import normalize as ON
from tests.util import synth_image


def _run(samples, layout, **kw):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    pipe = Pipeline(batch_size=len(samples), num_threads=3, device_id=None, prefetch_queue_depth=1)
    with pipe:
        x = fn.external_source(source=lambda: [np.ascontiguousarray(b) for b in samples], batch=True, layout=layout)
        pipe.set_outputs(fn.normalize(x, **kw))
    (out,) = pipe.run()
    assert pipe.executed_kernels() == ["host_normalize"]
    return [out.at(i) for i in range(len(samples))]


def _check(got, ref):
    for i, (g, r) in enumerate(zip(got, ref)):
        assert g.shape == r.shape, (i, g.shape, r.shape)
        assert np.allclose(g, r, rtol=1e-3, atol=1e-3), f"sample {i}: max err {np.abs(g - r).max()}"
        assert np.abs(g - r).max() < 2e-4 * max(1.0, np.abs(r).max())


CASES = [
    (dict(), None),
    (dict(axes=[0, 1]), (0, 1)),
    (dict(axis_names="HW"), (0, 1)),
    (dict(axes=[2]), (2,)),
    (dict(axes=[1, 2]), (1, 2)),
    (dict(axes=[0]), (0,)),
    (dict(axes=[0, 1], ddof=1, epsilon=0.25), (0, 1)),
    (dict(axes=[0, 1], mean=100.0), (0, 1)),
    (dict(axes=[0, 1], stddev=50.0, epsilon=0.5), (0, 1)),
    (dict(mean=3.0, stddev=2.0, scale=4.0, shift=-1.0), None),
]


@pytest.mark.parametrize("kw,axes", CASES)
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_per_sample_normalization_matches_numpy(kw, axes, dtype):
    rng = np.random.default_rng(12)
    shapes = [(37, 53, 3), (60, 80, 3), (1, 7, 3)]
    if dtype == np.uint8:
        batch = [synth_image(rng, h, w) for (h, w, _) in shapes]
    else:
        batch = [(rng.normal(5, 30, s) * np.linspace(0.5, 2, s[2])).astype(np.float32) for s in shapes]
    got = _run(batch, "HWC", **kw)
    okw = {"mean": kw.get("mean"), "stddev": kw.get("stddev"), "ddof": kw.get("ddof", 0), "eps": kw.get("epsilon", 0.0),
           "scale": kw.get("scale", 1.0), "shift": kw.get("shift", 0.0)}
    _check(got, [ON.normalize(b, axes, **okw).astype(np.float32) for b in batch])


def test_batch_normalization_and_constant_rows_and_integer_output():
    from dali_amd import types
    rng = np.random.default_rng(13)
    batch = [synth_image(rng, h, w) for (h, w) in [(40, 60), (100, 30), (64, 64)]]
    _check(_run(batch, "HWC", axes=[0, 1], batch=True, ddof=1),
           [r.astype(np.float32) for r in ON.normalize_batch(batch, (0, 1), ddof=1)])
    _check(_run(batch, "HWC", batch=True), [r.astype(np.float32) for r in ON.normalize_batch(batch, None)])
    spec = [rng.normal(-40, 12, (65, t)).astype(np.float32) for t in (31, 100)]
    spec[0][7, :] = 3.5                                         # zero variance -> zeros (ScaleRSqrtKeepZero)
    got = _run(spec, "ft", axes=[1])
    _check(got, [ON.normalize(b, (1,)).astype(np.float32) for b in spec])
    assert np.all(got[0][7] == 0)
    got = _run(batch[:1], "HWC", axes=[0, 1], scale=64.0, shift=128.0, dtype=types.UINT8)
    ref = np.clip(np.rint(ON.normalize(batch[0], (0, 1), scale=64.0, shift=128.0)), 0, 255)
    assert got[0].dtype == np.uint8 and np.abs(got[0].astype(np.int32) - ref.astype(np.int32)).max() <= 1
    with pytest.raises(RuntimeError, match="adjacent"):
        _run(batch[:1], "HWC", axes=[0, 2])
