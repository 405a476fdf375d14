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

import numpy as np


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


def run_nested(prior: Callable, loglike: Callable, ndim: int, nlive: int = 400, dlogz: float = 0.5,
               max_iter: int = 200000, max_calls: int = 5_000_000, batch: int = 1024, enlarge: float = 1.25, update_every: Optional[int] = None,
               seed: int = 0) -> NestedResult:
    """Nested sampling with vectorized callbacks.  Stops when the live points can add less than
    `dlogz` to ln Z (UltraNest's dlogz, evidence/ultranest/__init__.py:182), at `max_iter` replacements,
    or — so that a collapsing acceptance rate can never spin forever — once `max_calls` likelihood
    evaluations have been spent (the result then covers the iterations completed so far)."""
    rng = np.random.default_rng(seed)
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
