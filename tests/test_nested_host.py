"""CPU: the batched nested-sampling driver on the reference's Gaussian known-answer problems
(tests/test_polychord.py:75-151 with tests/test_examples/gaussian/*: unnormalised likelihood
-1/2 sum x^2 under Uniform(-10, 10) priors; analytic ln Z = ln(sqrt(2 pi)/20) per dimension)."""
import numpy as np
import pytest

from evidence_amd.nested import run_nested

LNZ_1D = float(np.log(np.sqrt(2 * np.pi) / 20.0))       # -2.0768


@pytest.mark.parametrize("ndim,nlive", [(1, 100), (2, 500)])
def test_gaussian_logz_known_answer(ndim, nlive):
    prior = lambda cube: -10.0 + 20.0 * cube                       # Uniform(-10, 10).ppf, vectorized
    loglike = lambda x: -0.5 * np.sum(x * x, axis=1)
    res = run_nested(prior, loglike, ndim, nlive=nlive, dlogz=0.05, seed=1)
    want = ndim * LNZ_1D
    assert abs(res.logz - want) < 0.5                               # the reference's own tolerance (:98, :139)
    assert abs(res.logz - want) < 4 * res.logzerr + 0.05
    w = np.exp(res.logwt)
    assert abs(w.sum() - 1) < 1e-9
    mean = (w[:, None] * res.samples).sum(axis=0)
    var = (w[:, None] * (res.samples - mean) ** 2).sum(axis=0)
    assert np.all(np.abs(mean) < 0.3) and np.all(np.abs(var - 1) < 0.35)
    assert res.ncall >= res.niter


def test_callbacks_are_called_with_batches():
    seen = []
    def loglike(x):
        seen.append(x.shape)
        return -0.5 * np.sum(x * x, axis=1)
    run_nested(lambda c: -10 + 20 * c, loglike, 2, nlive=50, dlogz=0.5, seed=2, batch=256)
    assert seen[0] == (50, 2) and all(len(s) == 2 and s[1] == 2 for s in seen)
    assert max(s[0] for s in seen) > 50


def test_max_calls_bounds_a_collapsing_run():
    # a likelihood spike the ellipsoid sampler cannot find efficiently: the call budget must end the run
    loglike = lambda x: -0.5 * np.sum(((x - 3.3) / 1e-4) ** 2, axis=1)
    res = run_nested(lambda c: -10 + 20 * c, loglike, 3, nlive=50, dlogz=0.01, seed=3, batch=64, max_calls=3000)
    assert res.ncall <= 3000 + 64 and np.isfinite(res.logz)


from evidence_amd.nested import run_nested_slice


@pytest.mark.parametrize("ndim,nlive", [(1, 200), (2, 400), (6, 400)])
def test_slice_sampler_gaussian_logz(ndim, nlive):
    prior = lambda cube: -10.0 + 20.0 * cube
    loglike = lambda x: -0.5 * np.sum(x * x, axis=1)
    res = run_nested_slice(prior, loglike, ndim, nlive=nlive, dlogz=0.05, seed=5)
    want = ndim * LNZ_1D
    assert abs(res.logz - want) < 0.5                      # tests/test_polychord.py:98,139 tolerance
    assert abs(res.logz - want) < 4 * res.logzerr + 0.1, (res.logz, want, res.logzerr)
    w = np.exp(res.logwt)
    mean = (w[:, None] * res.samples).sum(axis=0)
    var = (w[:, None] * (res.samples - mean) ** 2).sum(axis=0)
    assert np.all(np.abs(mean) < 0.35) and np.all(np.abs(var - 1) < 0.4)


def test_slice_sampler_bimodal_and_wrapped():
    # two well-separated modes of equal weight, and a circular parameter whose mode straddles the 0/1 seam
    def loglike(x):
        a = -0.5 * np.sum(((x - np.array([-5.0, 0.0])) / 0.5) ** 2, axis=1)
        b = -0.5 * np.sum(((x - np.array([5.0, 0.0])) / 0.5) ** 2, axis=1)
        return np.logaddexp(a, b)
    res = run_nested_slice(lambda c: -10 + 20 * c, loglike, 2, nlive=400, dlogz=0.05, seed=6)
    want = np.log(2 * 2 * np.pi * 0.25 / 400.0)          # two Gaussians of sigma 0.5 over a 20x20 prior
    assert abs(res.logz - want) < 0.5
    w = np.exp(res.logwt)
    left = w[res.samples[:, 0] < 0].sum()
    assert 0.3 < left < 0.7                               # both modes kept

    def loglike_circ(x):                                  # von-Mises-like bump centred on the seam of x0
        return 8.0 * np.cos(2 * np.pi * x[:, 0]) - 0.5 * ((x[:, 1] - 0.5) / 0.1) ** 2
    r1 = run_nested_slice(lambda c: c, loglike_circ, 2, nlive=300, dlogz=0.05, seed=7, wrapped=[True, False])
    from scipy.special import i0
    want = np.log(i0(8.0)) + np.log(np.sqrt(2 * np.pi) * 0.1)
    assert abs(r1.logz - want) < 0.5


def test_slice_sampler_respects_call_budget():
    res = run_nested_slice(lambda c: -10 + 20 * c, lambda x: -0.5 * np.sum(x * x, axis=1), 3, nlive=100,
                           dlogz=1e-6, seed=1, max_calls=20000)
    assert res.ncall < 20000 + 100 * 9 * 200 and np.isfinite(res.logz)


def test_fused_callback_gives_the_same_run():
    prior = lambda c: -10.0 + 20.0 * c
    loglike = lambda x: -0.5 * np.sum(x * x, axis=1)
    a = run_nested_slice(prior, loglike, 3, nlive=100, dlogz=0.5, seed=9)
    b = run_nested_slice(prior, loglike, 3, nlive=100, dlogz=0.5, seed=9,
                         prior_loglike=lambda c: (prior(c), loglike(prior(c))))
    assert a.logz == b.logz and a.ncall == b.ncall


class _HostLive:
    """A stand-in for GpuRVModel's resident live set (live_init / live_step / live_get / live_dead) that keeps the rows in
    numpy arrays and walks them with a deterministic host routine: the driver's `live=` path must be the `walker=` path,
    bit for bit, when both are fed the same walk — whatever order the driver does its bookkeeping in."""

    def __init__(self, prior, loglike, walk):
        self.prior, self.loglike, self.walk = prior, loglike, walk
        self.calls = []

    def live_init(self, cube):
        self.u = np.array(cube)
        self.theta = self.prior(self.u)
        self.logl = self.loglike(self.theta)
        self.dead_theta, self.dead_logl = [], []
        return self.logl.copy()

    def live_step(self, order, kdead, start, lstar, wrapped=None, nsteps=10, max_rounds=200, seed=0, chol=None):
        order, start = np.asarray(order), np.asarray(start)
        dead, alive = order[:kdead], order[kdead:]
        assert set(start) <= set(alive) and len(start) == kdead
        self.dead_theta.append(self.theta[dead].copy()); self.dead_logl.append(self.logl[dead].copy())
        if chol is None:                                  # "device" whitening: this stand-in factors the same matrix
            ua = self.u[alive]
            d0 = ua - ua.mean(axis=0)
            chol = np.linalg.cholesky(d0.T @ d0 / max(1, len(alive) - 1) + 1e-14 * np.eye(self.u.shape[1]))
        self.calls.append((kdead, float(lstar), int(seed)))
        wu, wt, wl, used = self.walk(self.u[start], self.theta[start], self.logl[start], lstar, chol, wrapped, nsteps, max_rounds, seed)
        self.u[dead], self.theta[dead], self.logl[dead] = wu, wt, wl
        return wl.copy(), used

    def live_get(self, cube=True, theta=True, logl=True):
        return (self.u.copy() if cube else None, self.theta.copy() if theta else None, self.logl.copy() if logl else None)

    def live_dead(self):
        return np.vstack(self.dead_theta), np.concatenate(self.dead_logl)


class _HostLiveSorting(_HostLive):
    """... with the ORDER kept by the stand-in as well (GpuRVModel.live_sort, round 4): the driver sends ranks among the
    survivors and never sees the order."""

    def live_sort(self, kdead):
        self.order = np.argsort(self.logl, kind="stable")
        self.sorted_for = kdead
        dl = self.logl[self.order[:kdead]]
        return dl.copy(), float(dl[-1]), float(self.logl[self.order[-1]])

    def live_step(self, order, kdead, start, lstar, *a, **k):
        assert order is None and self.sorted_for == kdead and lstar == self.logl[self.order[kdead - 1]]
        self.sorted_for = None
        return super().live_step(self.order, kdead, self.order[kdead:][np.asarray(start)], lstar, *a, **k)


@pytest.mark.parametrize("live_chol", ["device", "host"])
def test_resident_live_set_path_is_the_walker_path(live_chol):
    prior = lambda cube: -10.0 + 20.0 * cube
    loglike = lambda x: -0.5 * np.sum(x * x, axis=1)

    def walk(cube, theta, logl, lstar, chol, wrapped, nsteps, max_rounds, seed):
        # a crude but deterministic constrained move: shrink towards the centre of the cube until inside logL > lstar
        rng = np.random.default_rng(seed)
        c = cube.copy()
        used = 0
        for _ in range(nsteps):
            prop = np.clip(c + (rng.standard_normal(c.shape) @ chol.T) * 0.5, 0.0, np.nextafter(1.0, 0.0))
            ok = loglike(prior(prop)) > lstar
            used += len(c)
            c[ok] = prop[ok]
        th = prior(c)
        return c, th, loglike(th), used

    kw = dict(nlive=300, kbatch=100, nsteps=4, dlogz=0.05, max_calls=200_000, seed=9)
    ref = run_nested_slice(prior, loglike, 3, walker=walk, **kw)
    live = _HostLive(prior, loglike, walk)
    got = run_nested_slice(None, None, 3, live=live, live_chol=live_chol, **kw)
    assert got.niter == ref.niter and got.ncall == ref.ncall and got.logz == ref.logz and got.information == ref.information
    assert np.array_equal(got.samples, ref.samples) and np.array_equal(got.logl, ref.logl) and np.array_equal(got.logwt, ref.logwt)
    assert len(live.calls) == ref.niter // 100 and all(k == 100 for k, _, _ in live.calls)
    if live_chol == "device":
        # the order on the "device" too: the driver keeps no per-point state at all, and it is still the same run
        sorting = _HostLiveSorting(prior, loglike, walk)
        got2 = run_nested_slice(None, None, 3, live=sorting, **kw)
        assert got2.niter == ref.niter and got2.ncall == ref.ncall and got2.logz == ref.logz and got2.information == ref.information
        assert np.array_equal(got2.samples, ref.samples) and np.array_equal(got2.logl, ref.logl) and np.array_equal(got2.logwt, ref.logwt)
        assert sorting.calls == live.calls
    with pytest.raises(ValueError):
        run_nested_slice(None, None, 3, live=live, live_chol="elsewhere", **kw)
