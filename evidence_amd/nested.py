"""A minimal batched nested-sampling driver — the seam through which a sampler feeds the GPU.

The reference delegates sampling to PolyChord / UltraNest (third-party, not in the checkout,
not installed here) and only supplies `prior(cube)` and `loglike(theta)`
(evidence/ultranest/__init__.py:165-185).  This driver consumes callbacks with UltraNest's
`vectorized=True` signatures — prior((n, ndim)) -> (n, ndim), loglike((n, ndim)) -> (n,) — and
exists so that BASELINE.json configs[0] (400 live points through the callback boundary) and the
reference's Gaussian known-answer tests (tests/test_polychord.py:75-151: ln Z = -2.0768 in 1-D,
-4.1536 in 2-D) can run end to end.  It is deliberately simple: one bounding ellipsoid in the unit
cube, rejection sampling in batches, one replacement per iteration.  It is not a substitute for
UltraNest's region/step samplers on hard posteriors.
"""
from dataclasses import dataclass
from typing import Callable, Optional

import time

import numpy as np

from .settings import ultranest_defaults


@dataclass
class NestedResult:
    logz: float
    logzerr: float
    niter: int
    ncall: int
    information: float
    samples: np.ndarray          # dead + final live points (theta)
    logl: np.ndarray
    logwt: np.ndarray            # log posterior weights (normalised)
    timing: dict = None          # resident live set: seconds in the order step, waiting for the live step, and the loop turns


def _logaddexp_many(x):
    m = np.max(x)
    return m + np.log(np.sum(np.exp(x - m))) if np.isfinite(m) else m


class _Ellipsoid:
    """Bounding ellipsoid of the live points in the unit cube, enlarged."""

    def __init__(self, u, enlarge):
        self.ndim = u.shape[1]
        self.mean = u.mean(axis=0)
        d = u - self.mean
        cov = d.T @ d / max(1, u.shape[0] - 1) + 1e-12 * np.eye(self.ndim)
        self.chol = np.linalg.cholesky(cov)
        z = np.linalg.solve(self.chol, d.T)
        self.radius = np.sqrt(np.max(np.sum(z * z, axis=0))) * enlarge

    def sample(self, rng, n):
        z = rng.standard_normal((n, self.ndim))
        z *= (rng.random(n) ** (1.0 / self.ndim) / np.linalg.norm(z, axis=1))[:, None]
        return self.mean + self.radius * (z @ self.chol.T)


def run_nested(prior: Callable, loglike: Callable, ndim: int, nlive: Optional[int] = None, dlogz: float = 0.5,
               max_iter: int = 200000, max_calls: int = 5_000_000, batch: int = 1024, enlarge: float = 1.25, update_every: Optional[int] = None,
               seed: int = 0) -> NestedResult:
    """Nested sampling with vectorized callbacks.  Stops when the live points can add less than
    `dlogz` to ln Z (UltraNest's dlogz, evidence/ultranest/__init__.py:182), at `max_iter` replacements,
    or — so that a collapsing acceptance rate can never spin forever — once `max_calls` likelihood
    evaluations have been spent (the result then covers the iterations completed so far)."""
    rng = np.random.default_rng(seed)
    nlive = int(nlive or ultranest_defaults(ndim)["nlive"])          # the reference's default: 25 ndim
    u = rng.random((nlive, ndim))
    theta = np.asarray(prior(u), dtype=np.float64)
    logl = np.asarray(loglike(theta), dtype=np.float64)
    ncall = nlive
    update_every = update_every or max(1, nlive // 5)
    dead_theta, dead_logl, dead_logw = [], [], []
    logz, h, logx = -np.inf, 0.0, 0.0
    pool_u = pool_t = pool_l = None
    pos = 0
    it = 0
    while it < max_iter:
        worst = int(np.argmin(logl))
        lmin = logl[worst]
        logx_new = -(it + 1) / nlive
        logw = np.log(np.exp(logx) - np.exp(logx_new)) + lmin          # prior-mass shell x likelihood
        logz_new = np.logaddexp(logz, logw)
        # information H (Skilling 2006), updated incrementally
        h_old_term = np.exp(logz - logz_new) * (h + logz) if np.isfinite(logz) else 0.0
        h = np.exp(logw - logz_new) * lmin + h_old_term - logz_new
        logz, logx = logz_new, logx_new
        dead_theta.append(theta[worst].copy()); dead_logl.append(lmin); dead_logw.append(logw)
        # replacement: the first pooled candidate above the threshold.  A pool drawn from an older
        # (larger) ellipsoid stays valid — it is uniform on a superset of the constrained region.
        found = False
        while not found:
            if (pool_u is None or pos >= len(pool_u)) and ncall >= max_calls:
                break
            if pool_u is None or pos >= len(pool_u):
                cand = _Ellipsoid(u, enlarge).sample(rng, batch)
                cand = cand[np.all((cand >= 0.0) & (cand < 1.0), axis=1)]
                if len(cand) == 0:
                    continue
                pool_u = cand
                pool_t = np.asarray(prior(pool_u), dtype=np.float64)
                pool_l = np.asarray(loglike(pool_t), dtype=np.float64)      # one batch = one GPU launch
                ncall += len(pool_u)
                pos = 0
            while pos < len(pool_u):
                k = pos
                pos += 1
                if pool_l[k] > lmin:
                    u[worst], theta[worst], logl[worst] = pool_u[k], pool_t[k], pool_l[k]
                    found = True
                    break
        if not found:                      # budget exhausted: the point removed above stays dead, stop here
            logl[worst] = -np.inf
            keep = np.isfinite(logl)
            u, theta, logl = u[keep], theta[keep], logl[keep]
            it += 1
            break
        it += 1
        if it % update_every == 0:
            pool_u, pos = None, 0                                          # refresh the region now and then
        if np.max(logl) + logx < logz + np.log(np.expm1(dlogz)):           # remaining live mass is negligible
            break
    # final live points share the remaining prior mass
    logw_live = logx - np.log(max(1, len(logl))) + logl
    logz_final = np.logaddexp(logz, _logaddexp_many(logw_live))
    all_theta = np.vstack([np.array(dead_theta).reshape(-1, ndim), theta])
    all_logl = np.concatenate([dead_logl, logl])
    all_logw = np.concatenate([dead_logw, logw_live]) - logz_final
    return NestedResult(float(logz_final), float(np.sqrt(max(h, 0.0) / nlive)), it, ncall, float(h),
                        all_theta, all_logl, all_logw)


# --------------------------------------------------------------------------------------------------
# Batched nested slice sampling: the proposal scheme that keeps a GPU busy.
# --------------------------------------------------------------------------------------------------
def _chord(u, d, wrapped):
    """Range [tmin, tmax] (tmin < 0 < tmax) of t for which u + t d stays inside the unit cube; circular
    parameters have no walls and only limit |t d_i| to half a turn."""
    with np.errstate(divide="ignore", invalid="ignore"):
        t0 = (0.0 - u) / d
        t1 = (1.0 - u) / d
    lo = np.where(d != 0, np.minimum(t0, t1), -np.inf)
    hi = np.where(d != 0, np.maximum(t0, t1), np.inf)
    if wrapped is not None and wrapped.any():
        with np.errstate(divide="ignore"):
            half = np.where(d != 0, 0.5 / np.abs(d), np.inf)
        lo = np.where(wrapped, -half, lo)
        hi = np.where(wrapped, half, hi)
    return lo.max(axis=1), hi.min(axis=1)


_POOL = None


def _helper():
    """One helper thread for calls that block on the GPU while the host has arithmetic of its own to do."""
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=1, thread_name_prefix="rvll-live")
    return _POOL


def _stable_argsort(x):
    """np.argsort(x, kind="stable"), by way of the (vectorised, ~8x faster) unstable sort whenever that is provably
    the same permutation: no two equal neighbours in the sorted order means no ties to break."""
    order = np.argsort(x)
    xs = x[order]
    if np.any(xs[1:] == xs[:-1]) or np.isnan(xs[-1] if xs.size else 0.0):
        return np.argsort(x, kind="stable")
    return order


def run_nested_slice(prior: Callable, loglike: Callable, ndim: int, nlive: Optional[int] = None, kbatch: Optional[int] = None,
                     nsteps: Optional[int] = None, dlogz: float = 0.5, max_iter: int = 10_000_000,
                     max_calls: int = 50_000_000, wrapped=None, seed: int = 0,
                     prior_loglike: Optional[Callable] = None, walker: Optional[Callable] = None,
                     live=None, live_chol: str = "device") -> NestedResult:
    """Nested sampling with `kbatch` deaths per iteration and batched hit-and-run slice sampling.

    `prior_loglike(cubes) -> (theta, logl)`, if given, replaces the prior + loglike pair inside the loop
    (GpuRVModel.prior_loglike_batch: one upload, two launches, one download per round).
    `walker(cube, theta, logl, lstar, chol, wrapped, nsteps, max_rounds, seed) -> (cube, theta, logl, ncalls)`,
    if given, runs all `nsteps` moves of all replacement walkers in ONE call (GpuRVModel.slice_walk: the whole
    walk — directions, chords, candidates, prior transform, log-L, accept / shrink — stays on the GPU).
    `live`, if given (a GpuRVModel), keeps the LIVE SET ITSELF on the GPU (GpuRVModel.live_init / live_step: rvll_live_*):
    unit-cube rows, theta and log-L of the live points and of the points that died never leave HBM during the run; per
    iteration the host sends the sort order and the walkers' start rows and reads back the new log-L — what it needs
    for the sort and the evidence sum.  `prior` / `loglike` / `walker` are then unused.  live_chol="device" (default) takes
    the whitening from the surviving rows' covariance summed on the device; "host" mirrors the cube rows on the host
    and factors them exactly as the other paths do — slower, and bit-identical to `walker=model.slice_walk` for a seed
    (the equivalence test's mode).

    Each iteration removes the `kbatch` lowest live points in order (the live count shrinks nlive,
    nlive-1, ... while they die, as in dynamic nested sampling), then draws `kbatch` replacements above
    the highest removed likelihood: walkers start from random surviving live points and take `nsteps`
    slice moves along random directions whitened by the live-point covariance; a slice starts as the whole
    chord inside the unit cube (circular `wrapped` parameters wrap instead) and is shrunk towards the
    current position until a proposal is accepted.  Every shrink round evaluates ALL unfinished walkers in
    one vectorized callback call — one prior + log-L launch on the GPU.  Defaults follow the reference's
    UltraNest wrapper (evidence_amd/settings.py: nlive = 25 ndim, nsteps = 3 ndim, dlogz = 0.5;
    evidence/ultranest/__init__.py:333-338) and :159-163 (wrapped parameters)."""
    rng = np.random.default_rng(seed)
    defaults = ultranest_defaults(ndim)
    nlive = int(nlive or defaults["nlive"])
    kbatch = int(kbatch or max(1, nlive // 4))
    if not 1 <= kbatch < nlive:
        raise ValueError("need 1 <= kbatch < nlive")
    nsteps = int(nsteps or defaults["nsteps"])
    wrapped = None if wrapped is None else np.asarray(wrapped, dtype=bool)
    u = rng.random((nlive, ndim))
    if live is not None:
        if live_chol not in ("device", "host"):
            raise ValueError(live_chol)
        theta = None
        logl = live.live_init(u)
        if live_chol == "device":
            u = None                                   # no host mirror of the rows at all
    else:
        theta = np.asarray(prior(u), dtype=np.float64)
        logl = np.asarray(loglike(theta), dtype=np.float64)
    ncall = nlive
    dead_theta, dead_logl, dead_logw = [], [], []
    logz, h, logx = -np.inf, 0.0, 0.0
    it = 0
    # the resident live set with the ORDER on the device as well (GpuRVModel.live_sort, round 4): no per-point state on the host
    # at all — per iteration the log-L of the dying points comes down (the evidence sums need them), ranks go up
    device_order = live is not None and u is None and hasattr(live, "live_sort")
    top = float(np.max(logl)) if device_order else None
    timing = {"order_s": 0.0, "step_wait_s": 0.0, "turns": 0}
    while it < max_iter and ncall < max_calls:
        timing["turns"] += 1
        t_turn = time.perf_counter()
        if device_order:
            dl, lstar, top = live.live_sort(kbatch)
            timing["order_s"] += time.perf_counter() - t_turn
            ranks = rng.integers(0, nlive - kbatch, kbatch)          # (the draw the host order's alive[rng.integers(...)] makes)
            seed_it = int(rng.integers(0, 2 ** 62))
            pending = _helper().submit(live.live_step, None, kbatch, ranks, lstar, wrapped, nsteps, 200, seed_it)
            order = dead = None
        else:
            order = _stable_argsort(logl)
            dead = order[:kbatch]
            lstar = logl[dead[-1]]
            dl = logl[dead]
            timing["order_s"] += time.perf_counter() - t_turn
        if device_order:
            pass
        elif live is not None and u is None:
            # the resident live set, whitening on the device: the walk needs nothing of this iteration's evidence
            # bookkeeping, so it starts first — on a helper thread (the C call releases the interpreter lock) — and the
            # vectorised sums below run on the host while the GPU walks
            alive = order[kbatch:]
            start = alive[rng.integers(0, len(alive), kbatch)]
            seed_it = int(rng.integers(0, 2 ** 62))
            pending = _helper().submit(live.live_step, order, kbatch, start, lstar, wrapped, nsteps, 200, seed_it)
        else:
            pending = None
        # the kbatch deaths in order, live count nlive - i while they die — vectorised (this loop used to cost more
        # than the likelihood calls): X shrinks by exp(-1/(nlive - i)), w_i = (X_{i-1} - X_i) L_i, Z accumulates,
        # and the information H follows from A = sum_j w_j ln L_j / Z = H + ln Z
        logx_seq = logx - np.cumsum(1.0 / (nlive - np.arange(kbatch)))
        logx_prev = np.concatenate([[logx], logx_seq[:-1]])
        logw = logx_prev + np.log1p(-np.exp(logx_seq - logx_prev)) + dl
        logz_seq = np.logaddexp.accumulate(np.concatenate([[logz], logw]))[1:]
        big = max(logz, float(np.max(logw))) if np.isfinite(logz) else float(np.max(logw))
        a_prev = np.exp(logz - big) * (h + logz) if np.isfinite(logz) else 0.0
        a_last = np.exp(big - logz_seq[-1]) * (a_prev + float(np.sum(np.exp(logw - big) * dl)))
        logz, logx = float(logz_seq[-1]), float(logx_seq[-1])
        h = float(a_last - logz)
        if live is None:
            dead_theta.append(theta[dead])                                                # (index arrays: already copies)
        dead_logl.append(dl); dead_logw.append(logw)
        it += kbatch
        if pending is not None:
            t_wait = time.perf_counter()
            wl, used = pending.result()
            timing["step_wait_s"] += time.perf_counter() - t_wait
            ncall += int(used)
            if device_order:
                top = max(top, float(np.max(wl)))                     # the survivors' highest and the newcomers'
            else:
                logl[dead] = wl
                top = np.max(logl)
            if top + logx < logz + np.log(np.expm1(dlogz)):
                break
            continue
        alive = order[kbatch:]
        # whitening from the surviving live points
        chol = None
        if u is not None:
            ua = u[alive]
            d0 = ua - ua.mean(axis=0)
            cov = d0.T @ d0 / max(1, len(alive) - 1) + 1e-14 * np.eye(ndim)
            chol = np.linalg.cholesky(cov)
        start = alive[rng.integers(0, len(alive), kbatch)]
        if live is not None:
            # live_chol="host": order and start rows up, the new log-L of the replaced rows down, and the mirror of the rows
            wl, used = live.live_step(order, kbatch, start, lstar, wrapped, nsteps, 200, int(rng.integers(0, 2 ** 62)), chol=chol)
            ncall += int(used)
            logl[dead] = wl
            u = live.live_get()[0]
            if np.max(logl) + logx < logz + np.log(np.expm1(dlogz)):
                break
            continue
        wu, wt, wl = u[start], theta[start], logl[start]
        if walker is not None:
            wu, wt, wl, used = walker(wu, wt, wl, lstar, chol, wrapped, nsteps, 200, int(rng.integers(0, 2 ** 62)))
            ncall += int(used)
        for _ in range(0 if walker is not None else nsteps):
            z = rng.standard_normal((kbatch, ndim))
            d = z @ chol.T
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            tmin, tmax = _chord(wu, d, wrapped)
            todo = np.arange(kbatch)
            rounds = 0
            while todo.size and rounds < 200:
                t = tmin[todo] + (tmax[todo] - tmin[todo]) * rng.random(todo.size)
                cand = wu[todo] + t[:, None] * d[todo]
                if wrapped is not None:
                    cand[:, wrapped] %= 1.0
                cand = np.clip(cand, 0.0, np.nextafter(1.0, 0.0))
                if prior_loglike is not None:
                    ct, cl = prior_loglike(cand)                        # one round trip to the GPU
                else:
                    ct = np.asarray(prior(cand), dtype=np.float64)
                    cl = np.asarray(loglike(ct), dtype=np.float64)      # one batch = one GPU launch
                ncall += todo.size
                ok = cl > lstar
                acc = todo[ok]
                wu[acc], wt[acc], wl[acc] = cand[ok], ct[ok], cl[ok]
                rej = todo[~ok]
                neg = t[~ok] < 0                                        # shrink the bracket towards t = 0
                tmin[rej[neg]] = t[~ok][neg]
                tmax[rej[~neg]] = t[~ok][~neg]
                todo = rej
                rounds += 1
        u[dead], theta[dead], logl[dead] = wu, wt, wl
        if np.max(logl) + logx < logz + np.log(np.expm1(dlogz)):
            break
    if device_order:
        logl = live.live_get(cube=False, theta=False)[2]              # the live points' log-L: once, at the end
    logw_live = logx - np.log(nlive) + logl
    logz_final = np.logaddexp(logz, _logaddexp_many(logw_live))
    if live is not None and hasattr(live, "live_dead_count"):
        # the samples come down once, straight into the array that is returned: dead points first, then the live set
        ndead = live.live_dead_count()
        all_theta = np.empty((ndead + nlive, ndim))
        live.live_dead(theta_out=all_theta[:ndead])
        live.live_get(cube=False, logl=False, theta_out=all_theta[ndead:])
    else:
        if live is not None:
            dead_theta = [live.live_dead()[0]]
            theta = live.live_get()[1]
        all_theta = np.vstack([a.reshape(-1, ndim) for a in dead_theta] + [theta])
    all_logl = np.concatenate(dead_logl + [logl])
    all_logw = np.concatenate(dead_logw + [logw_live]) - logz_final
    return NestedResult(float(logz_final), float(np.sqrt(max(h, 0.0) / nlive)), it, ncall, float(h),
                        all_theta, all_logl, all_logw, timing)
