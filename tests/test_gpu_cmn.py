"""GPU parity: stand-alone CropMirrorNormalize vs the oracle -- bit-exact (same sub/mul order, same
fp16 ties-away rounding as the CPU backend)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu

DT = {"float32": (2, O.F32, np.uint32), "float16": (1, O.F16, np.uint16), "uint8": (0, O.U8, np.uint8),
      "int8": (3, O.I8, np.uint8)}


def _run(imgs, anchors, crop, **kw):
    from dali_amd import backend as B
    dev = [torch.from_numpy(np.ascontiguousarray(im)).cuda() for im in imgs]
    out = B.cmn_batch(dev, anchors, crop, **kw)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("dtype", list(DT))
@pytest.mark.parametrize("layout", ["CHW", "HWC"])
def test_cmn_matches_oracle(dtype, layout):
    rng = np.random.default_rng(3)
    kd, od, vt = DT[dtype]
    # reference test shapes: dali/test/python/operator_1/test_crop_mirror_normalize.py:980-1040
    shapes = [(10, 20, 3), (1, 24 * 128 + 1, 3), (1, 24 * 128 - 1, 3), (999, 999, 3), (224, 224, 3), (37, 53, 3)]
    mean, inv = O.cmn_norm_args([0.485 * 255, 0.456 * 255, 0.406 * 255], [0.229 * 255, 0.224 * 255, 0.225 * 255])
    for shp in shapes:
        img = (np.arange(np.prod(shp)) % 256).astype(np.uint8).reshape(shp)
        ch, cw = max(1, shp[0] * 3 // 4), max(1, shp[1] * 3 // 4)
        ay, ax = O.crop_anchor(0.5, ch, shp[0]), O.crop_anchor(0.5, cw, shp[1])
        for mirror in (0, 1):
            got = _run([img], [(ay, ax)], (ch, cw), mirror=[mirror], mean=mean, inv_std=inv, out_dtype=kd,
                       out_layout=1 if layout == "CHW" else 0)[0]
            ref = O.cmn_u8(img, (ay, ax), (ch, cw), mirror=bool(mirror), mean=mean, inv_std=inv, layout=layout,
                           dtype=od)
            assert np.array_equal(got.view(vt), ref.view(vt)), (shp, mirror)


def test_cmn_std1_identity_random_and_scalar_args():
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (61, 47, 3), dtype=np.uint8)
    # scalar mean/std, fp16 ties: values k + 0.5 ulp patterns come from mean = 0.5
    mean, inv = O.cmn_norm_args([0.5], [1.0])
    got = _run([img], [(3, 5)], (40, 30), mean=mean, inv_std=inv, out_dtype=1)[0]
    ref = O.cmn_u8(img, (3, 5), (40, 30), mean=mean, inv_std=inv, dtype=O.F16)
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16))
    # no normalisation at all: plain crop + layout change, any dtype
    got = _run([img], [(0, 0)], (61, 47), out_dtype=0, out_layout=0)[0]
    assert np.array_equal(got, img)


def test_cmn_pad_output_and_out_of_bounds_fill():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    mean, inv = O.cmn_norm_args([10, 20, 30], [2, 3, 4])
    got = _run([img], [(-4, -7)], (32, 48), mean=mean, inv_std=inv, fill_values=(1.0, 2.0, 3.0), out_dtype=2,
               pad_output=True, mirror=[1])[0]
    ref = O.cmn_u8(img, (-4, -7), (32, 48), mirror=True, mean=np.r_[mean, 0], inv_std=np.r_[inv, 0],
                   fill_values=(1.0, 2.0, 3.0), pad_output=True, pad_oob=True, dtype=O.F32)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_cmn_batch_of_different_inputs():
    rng = np.random.default_rng(6)
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in [(224, 224), (300, 250), (230, 500)]]
    anchors = [(0, 0), (30, 11), (3, 200)]
    mean, inv = O.cmn_norm_args([120, 121, 122], [60, 61, 62])
    got = _run(imgs, anchors, (224, 224), mean=mean, inv_std=inv, out_dtype=1, mirror=[0, 1, 0])
    for i, im in enumerate(imgs):
        ref = O.cmn_u8(im, anchors[i], (224, 224), mirror=bool([0, 1, 0][i]), mean=mean, inv_std=inv, dtype=O.F16)
        assert np.array_equal(got[i].view(np.uint16), ref.view(np.uint16))
