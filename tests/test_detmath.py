"""Accuracy of the deterministic elementary functions (csrc/detmath.h) against float64 libm.
They are shared by the HIP kernels and the oracle's det mode, so they are pinned independently here."""
import numpy as np


def _ulp(got, ref64):
    ref32 = ref64.astype(np.float32)
    u = np.spacing(np.maximum(np.abs(ref32), np.float32(1e-30))).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64) / u


def _vec(fn, *args):
    return np.array([fn(*[float(a) for a in t]) for t in zip(*args)], dtype=np.float32)


def test_exp(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-20, 0, 20000), rng.uniform(-87, 10, 5000), [0.0, -0.0, -1e-8, -87.2]]).astype(np.float32)
    got = _vec(L.orc_dm_expf, x)
    assert _ulp(got, np.exp(x.astype(np.float64))).max() <= 2.0
    assert L.orc_dm_expf(-100.0) == 0.0 and L.orc_dm_expf(0.0) == 1.0


def test_exp2(oracle):
    L = oracle.lib()
    x = np.random.default_rng(1).uniform(-3, 6, 20000).astype(np.float32)
    got = _vec(L.orc_dm_exp2f, x)
    assert _ulp(got, np.exp2(x.astype(np.float64))).max() <= 1.5
    for n in range(-5, 6):
        assert L.orc_dm_exp2f(float(n)) == 2.0 ** n


def test_atan2(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(2)
    y = rng.normal(0, 1, 30000).astype(np.float32)
    x = rng.normal(0, 1, 30000).astype(np.float32)
    got = _vec(L.orc_dm_atan2f, y, x)
    ref = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    assert np.abs(got - ref).max() < 4e-7
    assert L.orc_dm_atan2f(0.0, 0.0) == 0.0
    assert abs(L.orc_dm_atan2f(1.0, 0.0) - np.pi / 2) < 2e-7 and abs(L.orc_dm_atan2f(0.0, -1.0) - np.pi) < 3e-7
    assert abs(L.orc_dm_atan2f(-1.0, -1.0) + 3 * np.pi / 4) < 3e-7


def test_sincos(oracle):
    L = oracle.lib()
    t = np.random.default_rng(3).uniform(-0.5, 7.0, 30000).astype(np.float32)
    s = _vec(L.orc_dm_sinf, t)
    c = _vec(L.orc_dm_cosf, t)
    assert np.abs(s - np.sin(t.astype(np.float64))).max() < 2.5e-7
    assert np.abs(c - np.cos(t.astype(np.float64))).max() < 2.5e-7
    assert L.orc_dm_sinf(0.0) == 0.0 and L.orc_dm_cosf(0.0) == 1.0


def test_ceil_log2_exact(oracle):
    L = oracle.lib()
    for e in range(-20, 30):
        p = np.float32(2.0 ** e)
        assert L.orc_dm_ceil_log2f(float(p)) == e
        assert L.orc_dm_ceil_log2f(float(np.nextafter(p, np.float32(np.inf)))) == e + 1
        assert L.orc_dm_ceil_log2f(float(np.nextafter(p, np.float32(0)))) == e
    x = np.random.default_rng(4).uniform(1e-3, 1e6, 5000).astype(np.float32)
    got = np.array([L.orc_dm_ceil_log2f(float(v)) for v in x])
    assert np.array_equal(got, np.ceil(np.log2(x.astype(np.float64))).astype(int))


def test_div_2pi_matches_ieee_division(oracle):
    """the kernels' 3-operation x / (2 pi) (detmath.h: dm_div_2pi) against the correctly rounded float32 division it replaces;
    the full range [2^-103, 256] was checked exhaustively once (931 135 489 values, 0 mismatches) with the same C code"""
    L = oracle.lib()
    rng = np.random.default_rng(7)
    c = np.float32(2.0) * np.float32(np.pi)
    bits = np.concatenate([rng.integers(0x0c000000, 0x43800000, 60000, dtype=np.int64),        # random floats of the verified range
                           np.arange(0x40c90fdb - 2000, 0x40c90fdb + 2000, dtype=np.int64),     # around 2 pi
                           np.arange(0x0f800000 - 300, 0x0f800000 + 300, dtype=np.int64),       # around the 2^-96 guard
                           rng.integers(1, 0x0c000000, 3000, dtype=np.int64)])                  # below it, denormals included
    x = bits.astype(np.uint32).view(np.float32)
    x = np.concatenate([x, -x, np.array([0.0, -0.0, 8 * 6.2831855, 36 * 6.2831855], dtype=np.float32)])
    got = _vec(L.orc_dm_div_2pi, x)
    ref = x / c
    assert got.view(np.uint32).tolist() == ref.view(np.uint32).tolist()


def test_branch_free_exp_is_the_same_function(oracle):
    """dm_expf_nb (kernels: clamp + select) against dm_expf (oracle: early returns), bit for bit, range ends included"""
    L = oracle.lib()
    rng = np.random.default_rng(8)
    x = np.concatenate([rng.uniform(-100, 100, 40000), rng.uniform(-12, 0, 20000), [-87.3, -87.30001, -87.29999, 88.7, 88.70001, 0.0, -0.0, -1e30, 1e30]])
    x = x.astype(np.float32)
    a = _vec(L.orc_dm_expf, x)
    b = _vec(L.orc_dm_expf_nb, x)
    assert a.view(np.uint32).tolist() == b.view(np.uint32).tolist()


def test_single_scaling_exp_for_non_positive_arguments(oracle):
    """dm_expf_nb_nonpos (kernels, x <= 0: one power-of-two scaling) against dm_expf, bit for bit; the whole range
    [-87.3, -0] (1 118 738 843 floats) was compared once with the same C code, 0 mismatches"""
    L = oracle.lib()
    rng = np.random.default_rng(9)
    x = np.concatenate([-rng.uniform(0, 12, 40000), -rng.uniform(0, 100, 20000), [-87.3, -87.30001, -87.29999, -0.0, 0.0, -1e-30, -1e30]]).astype(np.float32)
    a = _vec(L.orc_dm_expf, x)
    b = _vec(L.orc_dm_expf_nb_nonpos, x)
    assert a.view(np.uint32).tolist() == b.view(np.uint32).tolist()
