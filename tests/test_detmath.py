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


def test_kernel_argument_ranges_against_float64(oracle):
    """The functions on the EXACT argument ranges the kernels feed them (features.hip / extrema.hip), against float64, with the bound
    DESIGN.md section 2.2 claims (<= 2 ulp):
      exp    descriptor weights exp(-0.125 (ox^2 + oy^2)) with |ox|, |oy| <= 2.5 sqrt(2): arguments in [-3.2, 0]; orientation weights
             exp(-d^2 / (2 (1.5 sigma)^2)) over the 3-lambda square window: [-9, 0] -> every float of [-4.5, 0] in steps + a dense sample to -9.5
      atan2  image gradients: any sign combination, +-0 components, denormal magnitudes (flat image regions), all eight octants and the
             axes; compared as angles (absolute error against the float32 spacing at pi), since the bin index is floor(angle * n / 2 pi)
      sincos keypoint orientations: [0, 2 pi], the half-bin grid (k + 0.5) pi / 36 the orientation stage emits, and 2 pi itself
      exp2   sigma = seed * 2^((s + ds) / S): arguments in [-0.5, 2.5]"""
    L = oracle.lib()
    rng = np.random.default_rng(11)
    # exp: a regular grid of 180 001 points over [-4.5, 0] (every 2.5e-5) + random floats down to -9.5 + the exact products the kernels form
    x = np.concatenate([np.linspace(-4.5, 0.0, 180001), -rng.uniform(0, 9.5, 60000), -0.125 * rng.uniform(0, 25.0, 20000), [-0.0, 0.0, -1e-38, -1e-45]]).astype(np.float32)
    for fn in (L.orc_dm_expf, L.orc_dm_expf_nb_nonpos):
        got = _vec(fn, x)
        assert _ulp(got, np.exp(x.astype(np.float64))).max() <= 2.0
    # atan2: octants x magnitudes from denormal to 1e3, and the axes with signed zeros
    mags = np.array([1e-45, 3e-42, 1e-39, 1.2e-38, 1e-30, 1e-12, 1e-6, 3e-3, 0.04, 0.5, 1.0, 7.0, 255.0, 1e3], dtype=np.float32)
    ys, xs = [], []
    for my in mags:
        for mx in mags:
            for sy in (1.0, -1.0):
                for sx in (1.0, -1.0):
                    ys.append(sy * my)
                    xs.append(sx * mx)
    for z in (0.0, -0.0):
        for m in mags:
            for s in (1.0, -1.0):
                ys += [z, s * m]
                xs += [s * m, z]
    ry, rx = rng.normal(0, 0.05, 60000), rng.normal(0, 0.05, 60000)          # gradient-sized values: half differences of texels in [0, 1]
    y = np.concatenate([np.array(ys), ry]).astype(np.float32)
    xx = np.concatenate([np.array(xs), rx]).astype(np.float32)
    got = _vec(L.orc_dm_atan2f, y, xx)
    ref = np.arctan2(y.astype(np.float64), xx.astype(np.float64))
    # +-0 and denormal inputs: the reference's own convention for atan2(+-0, -x) = +-pi is kept in magnitude; the sign of a zero
    # y is not (both ends of the branch cut land in the same histogram bin after the [0, 2 pi) wrap): compare modulo 2 pi
    d = np.abs(((got.astype(np.float64) - ref + np.pi) % (2 * np.pi)) - np.pi)
    assert d.max() <= 2.0 * np.spacing(np.float32(np.pi)), float(d.max())
    assert np.all(np.isfinite(got)) and np.all(np.abs(got) <= np.float32(np.pi) + np.spacing(np.float32(np.pi)))
    # sincos on [0, 2 pi] and on the emitted orientation grid
    t = np.concatenate([rng.uniform(0.0, 2 * np.pi, 60000), (np.arange(0, 72) + 0.5) * np.pi / 36, [0.0, 2 * np.pi, np.pi, np.pi / 2]]).astype(np.float32)
    s = _vec(L.orc_dm_sinf, t)
    c = _vec(L.orc_dm_cosf, t)
    # 2 ulp of values near 1: absolute 2.4e-7 (relative ulps are meaningless at the zeros of sin / cos, where the argument's own rounding decides)
    assert np.abs(s - np.sin(t.astype(np.float64))).max() <= 2.4e-7 and np.abs(c - np.cos(t.astype(np.float64))).max() <= 2.4e-7
    # exp2
    e = np.concatenate([np.linspace(-0.5, 2.5, 60001), rng.uniform(-0.5, 2.5, 20000)]).astype(np.float32)
    assert _ulp(_vec(L.orc_dm_exp2f, e), np.exp2(e.astype(np.float64))).max() <= 2.0


def test_three_operation_division_by_3_over_every_integer_valued_float(oracle):
    """dm_div_3 (detmath.h; the orientation kernel's histogram smoothing) against x / 3.f for every integer-valued float in [0, 2^32]"""
    assert oracle.lib().orc_check_div_3() == 0
