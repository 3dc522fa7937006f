"""Oracle pinning: Philox4x32-10 known answers, per-sample stream derivation, crop generator
properties, and agreement between the oracle (plain-C restatement of libstdc++'s distributions) and
the product host library (which draws through std:: distributions like the reference)."""
import numpy as np
import pytest

from oracle import oracle as O

# Random123 kat_vectors for philox4x32-10: (counter, key) -> output
KAT = [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


@pytest.mark.parametrize("ctr,key,expect", KAT)
def test_philox_known_answers(ctr, key, expect):
    assert list(O.philox_block(ctr, key)) == expect


def test_philox_state_mapping_and_skipahead():
    # State(key, sequence, offset): ctr[0] = offset >> 2, ctr[1] = sequence, phase = offset & 3 (philox.h:31-36)
    p = O.Philox(key=0xa4093822 | (0x299f31d0 << 32), ctr_hi=0x13198a2e | (0x03707344 << 32),
                 ctr_lo=0x243f6a88 | (0x85a308d3 << 32))
    assert [p.next() for _ in range(4)] == KAT[2][2]
    # skipahead(n) == n calls of next()
    a, b = O.Philox(7, 3, 0), O.Philox(7, 3, 0)
    seq = [a.next() for _ in range(40)]
    b.skipahead(13)
    assert b.next() == seq[13]
    b.skipahead(7)
    assert b.next() == seq[21]
    c = O.Philox(7, 3, 0)
    c.skipahead_sequence(5)
    d = O.Philox(7, 8, 0)
    assert c.next() == d.next()


def test_product_philox_matches_oracle():
    import ctypes as C
    from dali_amd import backend as B, _capi as capi
    st = B.philox_state(0x0123456789abcdef, ctr_hi=42, ctr_lo=0xffffffffffffffff, phase=2)
    out = np.zeros(23, np.uint32)
    capi.host().daliamdPhiloxGenerate(C.byref(st), out.ctypes.data_as(C.c_void_p), 23)
    ref = O.Philox(0x0123456789abcdef, 42, 0xffffffffffffffff, 2)
    assert [ref.next() for _ in range(23)] == list(out)
    # counter carried into the high word
    assert st.ctr[1] == 43
    buf = C.create_string_buffer(80)
    assert capi.host().daliamdPhiloxStateToString(C.byref(st), buf, 80) == 0
    # format of dali/core/random/philox.cc:75-96: Philox_<key>_<ctr_hi>:<ctr_lo>_<phase>
    assert buf.value.decode().startswith("Philox_0123456789ABCDEF_000000000000002B:")
    st2 = capi.PhiloxState()
    assert capi.host().daliamdPhiloxStateFromString(C.byref(st2), buf.value) == 0
    assert (st2.key, st2.ctr[0], st2.ctr[1], st2.phase) == (st.key, st.ctr[0], st.ctr[1], st.phase)


def test_crop_windows_product_equals_oracle():
    from dali_amd import backend as B
    rng = np.random.default_rng(0)
    shapes = np.concatenate([rng.integers(1, 1200, (3000, 2)), [[1, 1], [1, 5000], [5000, 1], [2, 2], [224, 224]]])
    for seed, it in [(1234, 0), (-5, 7), (2 ** 40 + 3, 123)]:
        master = B.philox_state(seed, ctr_hi=it * len(shapes))
        a, c = B.random_crop_batch(master, shapes)
        ao, co = O.rrc_batch(seed, it, shapes)
        assert np.array_equal(a, ao) and np.array_equal(c, co)
    # non-default ranges, single attempt (forces the fallback path often)
    master = B.philox_state(9)
    a, c = B.random_crop_batch(master, shapes, aspect=(0.5, 0.6), area=(0.9, 1.0), num_attempts=1)
    ao, co = O.rrc_batch(9, 0, shapes, aspect=(0.5, 0.6), area=(0.9, 1.0), num_attempts=1)
    assert np.array_equal(a, ao) and np.array_equal(c, co)


def test_crop_window_properties():
    """The reference tests RRC by property (dali/test/python/operator_2/test_random_resized_crop.py:29-90):
    window inside the image, area fraction and aspect ratio inside the requested ranges (up to rounding)."""
    rng = np.random.default_rng(1)
    shapes = rng.integers(64, 1000, (5000, 2))
    a, c = O.rrc_batch(77, 0, shapes)
    assert (a >= 0).all() and (a + c <= shapes).all() and (c >= 1).all()
    area = c[:, 0] * c[:, 1] / (shapes[:, 0] * shapes[:, 1])
    ratio = c[:, 1] / c[:, 0]
    eps = 0.05
    assert (area >= 0.08 * (1 - eps) - 2 / shapes.min()).all() and (area <= 1.0 + 1e-6).all()
    assert (ratio >= 3 / 4 * (1 - eps)).all() and (ratio <= 4 / 3 * (1 + eps)).all()
    # the distribution is not degenerate
    assert area.std() > 0.2 and ratio.std() > 0.1
    # different iterations / samples give different windows
    a2, c2 = O.rrc_batch(77, 1, shapes)
    assert (np.abs(a - a2).sum(1) > 0).mean() > 0.9


def test_coin_flip_product_equals_oracle_and_is_fair():
    from dali_amd import backend as B
    for seed, it, p in [(5, 0, 0.5), (5, 3, 0.25), (99, 1, 1.0), (99, 1, 0.0)]:
        got = B.coin_flip_batch(B.philox_state(seed, ctr_hi=it * 4096), 4096, p)
        ref = O.coin_flip_batch(seed, it, 4096, p)
        assert np.array_equal(got, ref)
        if p == 1.0:
            assert got.all()
        elif p > 0:
            assert abs(got.mean() - p) < 0.03
        else:
            assert got.mean() < 0.01


def test_crop_anchor_rounding():
    from dali_amd import _capi as capi
    # CropAttr::CalculateAnchor: round(0.5 * (in - crop)), half away from zero (crop_attr.cc:235-236)
    for crop, insz in [(224, 225), (224, 256), (7, 8), (1, 4), (224, 224), (100, 1001)]:
        for norm in (0.0, 0.5, 1.0, 0.3):
            for rounding in ("round", "truncate"):
                exp = float(np.float32(norm)) * (insz - crop)
                want = int(np.floor(exp + 0.5)) if rounding == "round" else int(exp)
                assert O.crop_anchor(norm, crop, insz, rounding) == want
                assert capi.host().daliamdCropAnchor(norm, crop, insz, 1 if rounding == "round" else 0) == want
