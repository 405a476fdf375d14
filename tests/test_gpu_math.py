"""GPU: the device math the kernels are built from, checked in isolation through rvll_debug_eval."""
import numpy as np
import pytest

from evidence_amd import GpuRVModel
from evidence_amd.synthetic import make_workload

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180)]
ULP1 = 2.0 ** -53      # half-ulp of values in [1, 2) == one ulp of values in [0.5, 1)


@pytest.fixture(scope="module")
def dev(gpu_required):
    w = make_workload(1)
    m = GpuRVModel(w.fixedpardict, w.table, w.parnames)
    yield m
    m.close()


def test_div_exact_is_bit_identical_to_ieee_division(dev):
    """The Newton step E - f/f' must round exactly like the reference's division (trueanomaly.c:29)."""
    rng = np.random.default_rng(0)
    n = 4_000_000
    num = rng.normal(0, 1, n) * 10.0 ** rng.integers(-8, 3, n)       # f = E - e sin E - M
    den = rng.uniform(0.01, 1.99, n)                                 # 1 - e cos E with e <= 0.99
    mine = dev.debug_eval(2, num, den)
    ieee = dev.debug_eval(3, num, den)
    assert np.array_equal(mine, ieee)
    assert np.array_equal(ieee, num / den)                           # and both equal the host's correctly rounded quotient


def test_div_fast_within_2ulp(dev):
    rng = np.random.default_rng(1)
    num = rng.normal(0, 50, 1_000_000)
    den = rng.uniform(0.01, 3000.0, 1_000_000)
    got = dev.debug_eval(4, num, den)
    ref = num / den
    assert np.max(np.abs(got - ref) / np.abs(ref)) <= 2.5 * 2.0 ** -52


def test_sincos_accuracy_over_the_unreduced_mean_anomaly_range(dev):
    """|M| reaches ~1e4 rad on this path (never range-reduced, rvmodel:459)."""
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-2.0e4, 2.0e4, 2_000_000), rng.uniform(-7, 7, 500_000),
                        np.round(rng.uniform(-2.0e4, 2.0e4, 200_000) / (np.pi / 2)) * (np.pi / 2)])
    xl = x.astype(np.longdouble)
    s, c = dev.debug_eval(0, x), dev.debug_eval(1, x)
    es = np.max(np.abs(s.astype(np.longdouble) - np.sin(xl)))
    ec = np.max(np.abs(c.astype(np.longdouble) - np.cos(xl)))
    assert es <= 1.5 * ULP1 and ec <= 1.5 * ULP1, (float(es / ULP1), float(ec / ULP1))


def test_rotate_small_matches_direct(dev):
    rng = np.random.default_rng(3)
    x = rng.uniform(-10, 10, 500_000)
    h = rng.uniform(-1e-3, 1e-3, 500_000)
    xl, hl = x.astype(np.longdouble), h.astype(np.longdouble)
    s, c = dev.debug_eval(8, x, h), dev.debug_eval(9, x, h)
    ref_s = np.sin(xl) * np.cos(hl) + np.cos(xl) * np.sin(hl)
    ref_c = np.cos(xl) * np.cos(hl) - np.sin(xl) * np.sin(hl)
    assert np.max(np.abs(s.astype(np.longdouble) - ref_s)) <= 2.5 * ULP1
    assert np.max(np.abs(c.astype(np.longdouble) - ref_c)) <= 2.5 * ULP1


def test_log_pos_accuracy(dev):
    rng = np.random.default_rng(4)
    v = np.concatenate([10.0 ** rng.uniform(-6, 8, 1_000_000), rng.uniform(0.25, 2500.0, 1_000_000),
                        [1.0, 0.5, 2.0, 0.7071067811865476, 1.4142135623730951, 5e-324, 1e-310, 1e308]])
    got = dev.debug_eval(5, v)
    ref = np.log(v.astype(np.longdouble))
    err = np.abs(got.astype(np.longdouble) - ref) / np.maximum(np.abs(ref), np.longdouble(1e-300))
    absr = np.abs(got.astype(np.longdouble) - ref)
    assert np.max(np.minimum(err, absr)) <= 2.5 * 2.0 ** -52
    special = dev.debug_eval(5, np.array([0.0, -1.0, np.inf, np.nan]))
    assert special[0] == -np.inf and np.isnan(special[1]) and special[2] == np.inf and np.isnan(special[3])


def test_ndtri_against_scipy(dev):
    from scipy.special import ndtri
    rng = np.random.default_rng(5)
    p = np.concatenate([rng.random(500_000), 10.0 ** rng.uniform(-300, -1, 100_000), 1 - 10.0 ** rng.uniform(-15, -1, 100_000),
                        [0.0, 1.0, 0.5, 0.075, 0.925]])
    got = dev.debug_eval(7, p)
    ref = ndtri(p)
    fin = np.isfinite(ref)
    assert np.array_equal(got[~fin], ref[~fin])
    err = np.abs(got[fin] - ref[fin]) / np.maximum(np.abs(ref[fin]), 1e-300)
    err = np.where(ref[fin] == 0, np.abs(got[fin]), err)
    assert err.max() <= 1e-13, float(err.max())


def test_lane0_tree_without_lds_crossbar_is_the_shuffle_tree(dev):
    """The per-point reduction ends in a 6-step tree over the wave.  The kernels take lane 0's value from a
    v_permlane32_swap / v_permlane16_swap / row_shl-DPP form of it; it must be the shuffle tree bit for bit, and
    both must be the tree as written: v[i] += v[i + off] for off = 32, 16, 8, 4, 2, 1."""
    rng = np.random.default_rng(12)
    n = 256 * 512
    x = rng.normal(0, 1, n) * 10.0 ** rng.integers(-12, 12, n)
    shfl = dev.debug_eval(12, x)[::64]
    fast = dev.debug_eval(13, x)[::64]
    v = x.reshape(-1, 64).copy()
    for off in (32, 16, 8, 4, 2, 1):
        v[:, :off] = v[:, :off] + v[:, off:2 * off]
    assert np.array_equal(shfl, v[:, 0])
    assert np.array_equal(fast, v[:, 0])


def test_device_sincos_cr_is_correctly_rounded(dev):
    """The redo pass's sin / cos pair ON THE DEVICE (debug_eval 20 / 21: rvll_math.h sincos_cr — double-double reduction, table
    route with Ziv's test, full series where it declines) against mpmath at 200 bits: every value THE nearest double, from
    1e-300 to 1e300, next to multiples of pi/2 included (the host compilation of the same header: tests/test_hostmath.py)."""
    import math
    import mpmath
    mpmath.mp.prec = 200
    rng = np.random.default_rng(21)
    x = np.concatenate([rng.uniform(-20, 20, 4000), 10.0 ** rng.uniform(-300, 300, 1000) * rng.choice([-1, 1], 1000),
                        rng.uniform(1e9, 2e22, 3000), np.arange(1, 300) * (math.pi / 2), np.arange(0, 52) / 64.0,
                        np.array([0.0, 0.7853981633974483, -0.7853981633974483, 1e22, 2.0 ** 50, 2.0 ** 1023])])
    s, c = dev.debug_eval(20, x), dev.debug_eval(21, x)
    want_s = np.array([float(mpmath.sin(mpmath.mpf(float(v)))) for v in x])
    want_c = np.array([float(mpmath.cos(mpmath.mpf(float(v)))) for v in x])
    assert np.array_equal(s, want_s) and np.array_equal(c, want_c)
