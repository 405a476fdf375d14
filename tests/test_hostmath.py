"""CPU: the kernels' host+device math headers compiled for the host (tests/native/hostmath.hip) against
the golden prior vectors and scipy.  Needs hipcc (present in the build container and on the GPU box)."""
import ctypes as C
import math
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

import golden
import prior_cases as pc

HERE = Path(__file__).resolve().parent
SRC = HERE / "native" / "hostmath.hip"
LIB = HERE / "native" / "libhostmath.so"
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
dp = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def hm():
    if not Path(HIPCC).exists():
        pytest.skip("hipcc not available")
    hdrs = list((HERE.parent / "evidence_amd" / "csrc").glob("*.h"))
    if not LIB.exists() or LIB.stat().st_mtime < max(p.stat().st_mtime for p in [SRC] + hdrs):
        subprocess.run([HIPCC, "-O2", "-ffp-contract=off", "-mfma", "-fPIC", "-shared", "--offload-arch=gfx950",
                        f"-I{HERE.parent / 'evidence_amd' / 'csrc'}", str(SRC), "-o", str(LIB)], check=True)
    return C.CDLL(str(LIB))


def call(lib, fn, q, *args):
    q = np.ascontiguousarray(q, dtype=float)
    out = np.empty_like(q)
    getattr(lib, fn)(q.ctypes.data_as(dp), C.c_long(q.size), *[C.c_double(a) for a in args], out.ctypes.data_as(dp))
    return out


def test_special_function_priors_match_golden(hm):
    q, sets = golden.prior_sets()
    seen = set()
    for name, args, vals, raised in sets:
        if name == "Beta":
            got = call(hm, "hm_beta_ppf", q, args[0], args[1], math.lgamma(args[0]) + math.lgamma(args[1]) - math.lgamma(args[0] + args[1]))
        elif name == "Gamma":
            got = call(hm, "hm_gamma_ppf", q, args[0], args[1], math.lgamma(args[0]))
        elif name == "Alpha":
            got = call(hm, "hm_alpha_ppf", q, args[0], 0.5 * math.erfc(-args[0] / math.sqrt(2)))
        elif name == "Normal":
            got = call(hm, "hm_ndtri", q) * args[1] + args[0]
        else:
            continue
        seen.add(name)
        m = pc.comparable_mask(name, q, raised)
        err = pc.rel_err(got[m], vals[m])
        assert err.max() <= pc.base_tol(name, args), (name, args, float(err.max()), float(q[m][err.argmax()]))
    assert seen == {"Beta", "Gamma", "Alpha", "Normal"}


def test_beta_gamma_inverse_against_scipy_dense(hm):
    from scipy import special as sp
    rng = np.random.default_rng(0)
    q = np.concatenate([rng.random(4000), 10.0 ** rng.uniform(-12, -1, 500), 1 - 10.0 ** rng.uniform(-12, -1, 500)])
    for a, b in [(0.867, 3.03), (2, 5), (0.5, 0.5), (12, 1.5), (1, 1), (50, 80), (300, 2)]:
        got = call(hm, "hm_beta_ppf", q, a, b, float(sp.betaln(a, b)))
        assert pc.rel_err(got, sp.betaincinv(a, b, q)).max() <= 5e-14, (a, b)
        back = call(hm, "hm_betainc", got, a, b, float(sp.betaln(a, b)))           # round trip: I_x(ppf(q)) = q
        mid = q <= 0.99          # x -> 1 is ill-conditioned for b < 1 (infinite slope): round trip only away from it
        assert pc.rel_err(back[mid], q[mid]).max() <= 1e-12, (a, b)
    for al, be in [(2, 3), (0.7, 0.1), (25, 2), (1, 1), (5, 0.5)]:
        got = call(hm, "hm_gamma_ppf", q, al, be, math.lgamma(al))
        assert pc.rel_err(got, sp.gammaincinv(al, q) / be).max() <= 5e-14, (al, be)


def test_sincos_host_build_accuracy(hm):
    rng = np.random.default_rng(1)
    x = rng.uniform(-2e4, 2e4, 200000)
    s, c = np.empty_like(x), np.empty_like(x)
    hm.hm_sincos(x.ctypes.data_as(dp), C.c_long(x.size), s.ctypes.data_as(dp), c.ctypes.data_as(dp))
    xl = x.astype(np.longdouble)
    assert np.max(np.abs(s.astype(np.longdouble) - np.sin(xl))) <= 1.5 * 2.0 ** -53
    assert np.max(np.abs(c.astype(np.longdouble) - np.cos(xl))) <= 1.5 * 2.0 ** -53


def test_tabulated_quantiles_equal_the_full_solver(hm):
    """The device tabulates each Beta/Gamma prior once.  Cubic start + one Newton polish lands on the full
    solver's answer; the quintic interpolant, whose error against the full solver is MEASURED when the table is
    built, replaces the polish when that error is below the tolerance — both forms are checked here."""
    from scipy import special as sp
    n = hm.hm_table_n()
    hm.hm_beta_table.restype = C.c_double
    hm.hm_gamma_table.restype = C.c_double
    hm.hm_table_direct_tol.restype = C.c_double
    tol = hm.hm_table_direct_tol()
    rng = np.random.default_rng(2)
    q = np.concatenate([rng.random(20000), 10.0 ** rng.uniform(-12, -1, 2000), 1 - 10.0 ** rng.uniform(-12, -1, 2000),
                        [0.0, 1.0, 1e-14, 1 - 1e-15, 0.5]])
    mid = (q >= 1e-12) & (q <= 1 - 1e-12)
    report = []
    for a, b in [(0.867, 3.03), (2, 5), (12, 1.5), (50, 80), (0.1, 0.2), (0.5, 0.5), (1.0, 1.0), (300.0, 2.0)]:
        lb = float(sp.betaln(a, b))
        z, dz, out = np.empty(n), np.empty(2 * n), np.empty_like(q)
        err = hm.hm_beta_table(C.c_double(a), C.c_double(b), C.c_double(lb), z.ctypes.data_as(dp), dz.ctypes.data_as(dp))
        assert np.isfinite(z).all() and np.isfinite(dz).all() and np.all(np.diff(z) > 0)
        full = call(hm, "hm_beta_ppf", q, a, b, lb)
        for direct in (0, 1):
            hm.hm_beta_ppf_table(q.ctypes.data_as(dp), C.c_long(q.size), C.c_double(a), C.c_double(b), C.c_double(lb),
                                 z.ctypes.data_as(dp), dz.ctypes.data_as(dp), C.c_int(direct), out.ctypes.data_as(dp))
            assert out[-5] == 0.0 and out[-4] == 1.0
            worst = pc.rel_err(out[mid], full[mid]).max()
            if direct == 0:
                assert worst <= 1e-13, (a, b)
            else:
                report.append(("beta", a, b, err, worst))
                # the measured table error bounds what interpolation alone delivers (z = logit x: d ln x = (1-x) dz)
                assert worst <= max(4 * err, 2e-15) + 1e-15, (a, b, err, worst)
                if err <= tol:
                    assert worst <= 1e-13, (a, b, err, worst)
    for al, be in [(2, 3), (0.7, 0.1), (25, 2), (0.05, 1.0), (1.0, 1.0)]:
        lg = math.lgamma(al)
        z, dz, out = np.empty(n), np.empty(2 * n), np.empty_like(q)
        err = hm.hm_gamma_table(C.c_double(al), C.c_double(lg), z.ctypes.data_as(dp), dz.ctypes.data_as(dp))
        full = call(hm, "hm_gamma_ppf", q, al, be, lg)
        for direct in (0, 1):
            hm.hm_gamma_ppf_table(q.ctypes.data_as(dp), C.c_long(q.size), C.c_double(al), C.c_double(be), C.c_double(lg),
                                  z.ctypes.data_as(dp), dz.ctypes.data_as(dp), C.c_int(direct), out.ctypes.data_as(dp))
            worst = pc.rel_err(out[mid], full[mid]).max()
            if direct == 0:
                assert worst <= 1e-13, (al, be)
            else:
                report.append(("gamma", al, be, err, worst))
                assert worst <= max(4 * err, 2e-15) + 1e-15, (al, be, err, worst)
                if err <= tol:
                    assert worst <= 1e-13, (al, be, err, worst)
    print(report)
    # the shapes the shipped configs use must qualify for the interpolation-only path
    assert [r[3] <= tol for r in report if r[:3] in (("beta", 0.867, 3.03), ("beta", 2, 5))] == [True, True]


def test_sincos_of_any_finite_double(hm):
    """sincos_any (rvll_math.h): the short reduction up to 2^50, the long (Payne - Hanek) one beyond, against glibc, which
    reduces every argument exactly: <= 1.5 ulp of 1 over random arguments up to 1e300, at exact powers of two, around the
    switch at 2^50, and at the doubles closest to multiples of pi/2 (where a short reduction loses everything) with their
    RELATIVE accuracy kept; the long reduction by itself is also checked from 1 upwards, where the short one is valid too."""
    rng = np.random.default_rng(12)
    x = np.concatenate([
        rng.uniform(-1, 1, 20000) * 10.0 ** rng.uniform(0, 300, 20000),
        rng.uniform(-1, 1, 20000) * 2.0 ** rng.uniform(28, 56, 20000),
        2.0 ** np.arange(0, 1024), -(2.0 ** np.arange(0, 1024)),
        2.0 ** 50 + np.arange(-64, 65) * 0.25,
        np.array([6381956970095103.0 * 2.0 ** 797, 5319372648326541416707072.0, 1e22, 2.343e22, -1.03e18]),
        rng.integers(1, 2 ** 22, 5000) * np.pi,                      # the doubles next to multiples of pi: sin ~ 1e-10
        rng.integers(1, 2 ** 22, 5000) * np.pi + np.pi / 2,          # ... and cos
    ])
    rs = np.array([math.sin(v) for v in x])
    rc = np.array([math.cos(v) for v in x])
    ulp = 2.0 ** -52
    for fn, sel in (("hm_sincos_any", np.ones(x.size, bool)), ("hm_sincos_long", np.abs(x) >= 1.0)):
        xs = np.ascontiguousarray(x[sel])
        s, c = np.empty_like(xs), np.empty_like(xs)
        getattr(hm, fn)(xs.ctypes.data_as(dp), C.c_long(xs.size), s.ctypes.data_as(dp), c.ctypes.data_as(dp))
        es, ec = np.abs(s - rs[sel]), np.abs(c - rc[sel])
        assert es.max() <= 1.5 * ulp, (fn, float(es.max() / ulp), float(xs[es.argmax()]))
        assert ec.max() <= 1.5 * ulp, (fn, float(ec.max() / ulp), float(xs[ec.argmax()]))
        if fn == "hm_sincos_long":                          # the long reduction keeps small results RELATIVELY accurate
            tiny = np.abs(rs[sel]) < 1e-6                   # (the short one is absolute: 2^-60 per quadrant passed)
            assert tiny.sum() > 100
            assert np.max(es[tiny] / np.abs(rs[sel][tiny])) <= 4 * ulp
    bad = np.array([np.inf, -np.inf, np.nan])
    sb, cb = np.empty(3), np.empty(3)
    hm.hm_sincos_any(bad.ctypes.data_as(dp), C.c_long(3), sb.ctypes.data_as(dp), cb.ctypes.data_as(dp))
    assert np.isnan(sb).all() and np.isnan(cb).all()


def test_sincos_cr_short_route_is_correctly_rounded_where_it_answers(hm):
    """The table route of sincos_cr (rvll_math.h, sincos_dd_table): wherever it answers, both values are THE nearest doubles
    (mpmath at 200 bits), and it declines — Ziv's test; those go through the full series — for well under 1 % of the arguments."""
    import mpmath
    mpmath.mp.prec = 200
    rng = np.random.default_rng(12)
    x = np.concatenate([rng.uniform(-7, 7, 12000), rng.uniform(-0.8, 0.8, 6000), 10.0 ** rng.uniform(-30, 30, 3000) * rng.choice([-1, 1], 3000),
                        rng.uniform(1e9, 2e22, 6000), np.arange(0, 52) / 64.0, np.arange(0, 52) / 64.0 + 2.0 ** -60,
                        np.arange(1, 400) * (math.pi / 2), np.arange(1, 200) * (math.pi / 4),
                        rng.uniform(2.0 ** 45, 2.0 ** 47, 3000), rng.uniform(2.0 ** 30, 2.0 ** 46, 3000),                               # either side of where the reduction changes routes
                        (rng.integers(1, 2 ** 45, 3000) * (math.pi / 2)).astype(float)])        # next to large multiples of pi/2
    s, c, ok = np.empty_like(x), np.empty_like(x), np.empty(x.size, dtype=np.int32)
    hm.hm_sincos_cr_table(x.ctypes.data_as(dp), C.c_long(x.size), s.ctypes.data_as(dp), c.ctypes.data_as(dp),
                          ok.ctypes.data_as(C.POINTER(C.c_int)))
    want_s = np.array([float(mpmath.sin(mpmath.mpf(float(v)))) for v in x])
    want_c = np.array([float(mpmath.cos(mpmath.mpf(float(v)))) for v in x])
    yes = ok != 0
    assert yes.mean() > 0.99, float(yes.mean())
    assert np.array_equal(s[yes], want_s[yes]) and np.array_equal(c[yes], want_c[yes])
    # and the routine as a whole (short route, full series where it declines)
    hm.hm_sincos_cr(x.ctypes.data_as(dp), C.c_long(x.size), s.ctypes.data_as(dp), c.ctypes.data_as(dp))
    assert np.array_equal(s, want_s) and np.array_equal(c, want_c)


def test_sincos_cr_is_correctly_rounded(hm):
    """sincos_cr (rvll_math.h, round 4): double-double reduction + Taylor series, rounded once — against mpmath at 200 bits on
    arguments from 1e-300 to 1e300, next to multiples of pi/2 included: every result is THE nearest double (0 ulp off); and
    how often glibc's sin / cos are not (that share is what a correctly rounded device routine can still differ by)."""
    import mpmath
    mpmath.mp.prec = 200
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(-20, 20, 3000), 10.0 ** rng.uniform(-300, 300, 1500) * rng.choice([-1, 1], 1500),
                        rng.uniform(1e9, 2e22, 1500), np.arange(1, 400) * (math.pi / 2), np.arange(1, 400) * (math.pi / 2) * (1 + 2.0 ** -40),
                        np.array([0.0, 0.7853981633974483, -0.7853981633974483, 1e22, 2.0 ** 50, 2.0 ** 1023])])
    s, c = np.empty_like(x), np.empty_like(x)
    hm.hm_sincos_cr(x.ctypes.data_as(dp), C.c_long(x.size), s.ctypes.data_as(dp), c.ctypes.data_as(dp))
    want_s = np.array([float(mpmath.sin(mpmath.mpf(float(v)))) for v in x])       # float(): round to nearest
    want_c = np.array([float(mpmath.cos(mpmath.mpf(float(v)))) for v in x])
    assert np.array_equal(s, want_s) and np.array_equal(c, want_c)
    glibc_off = np.mean((np.sin(x) != want_s) | (np.cos(x) != want_c))
    assert glibc_off < 0.02                                 # glibc itself: correctly rounded nearly always (it is what the oracle calls)
