"""The drop-in boundary: `prior(cube)` / `loglike(theta)` closures with exactly the
signatures and return conventions the reference hands to its samplers.

  make_polychord_callbacks  evidence/polychord/__init__.py:130-171
        prior(hypercube) -> new 1-D float64 array theta (sorted-prior groups transformed together)
        loglike(x)       -> (float logL, [])          # second element = derived parameters, nderived = 0
        call site: run_polychord(loglike, ndim, nderived, settings, prior)          (:190)
  make_ultranest_callbacks  evidence/ultranest/__init__.py:125-146
        prior(hypercube) -> new 1-D float64 array theta
        loglike(x)       -> float
        call site: ReactiveNestedSampler(parnames, loglike, prior, ..., wrapped_params=...)  (:165-172)
        with vectorized=True (the switch the reference left commented at :171) the same closures
        accept (n, ndim) arrays and return (n, ndim) / (n,) — the form that feeds the GPU.

Everything evaluates on the GPU through GpuRVModel; nothing here computes a likelihood.
"""
from typing import Callable, Sequence, Tuple

import numpy as np


def wrapped_params(parnames: Sequence[str]) -> np.ndarray:
    """Circular parameters, as evidence/ultranest/__init__.py:159-163 flags them."""
    return np.array([("omega" in p) or ("ml0" in p) for p in parnames], dtype=bool)


def make_polychord_callbacks(model, low_latency: bool = False) -> Tuple[Callable, Callable, int, int]:
    """(prior, loglike, ndim, nderived) for pypolychord.run_polychord.

    PolyChord calls prior(cube) and loglike(theta) one point at a time.  low_latency=True answers both through
    the model's persistent scalar-call kernel (GpuRVModel.scalar_server): ~19 us per prior + loglike pair at
    cfg3 instead of ~45 us with a launch and a synchronisation per call; same bits.  PolyChord's loglike(theta) always
    follows prior(cube) with the theta that call returned (evidence/polychord/__init__.py:130-171 hands it both
    closures): in the low-latency form prior() therefore asks for the pair in one request and loglike() answers from
    that when it is handed exactly that theta (compared element by element; anything else is evaluated as usual)."""
    ndim, nderived = len(model.parnames), 0
    if low_latency:
        model.scalar_server(True)
    last = {"theta": None, "logl": 0.0}

    def prior(hypercube):
        cube = np.asarray(hypercube, dtype=np.float64)
        if not low_latency:
            return model.prior_transform(cube)
        theta, logl = model.prior_loglike(cube)
        last["theta"], last["logl"] = theta.copy(), logl
        return theta

    def loglike(x):
        t = last["theta"]
        if t is not None:
            xa = np.asarray(x, dtype=np.float64)
            if xa.shape == t.shape and np.array_equal(xa, t):
                return (last["logl"], [])
        return (model.log_likelihood(x), [])

    return prior, loglike, ndim, nderived


def make_ultranest_callbacks(model, vectorized: bool = False, low_latency: bool = False, paired: bool = False) -> Tuple[Callable, Callable]:
    """(prior, loglike) for ultranest.ReactiveNestedSampler(parnames, loglike, prior, vectorized=...).
    low_latency (scalar form only): as in make_polychord_callbacks.
    paired (vectorized form): UltraNest evaluates loglike(p) on exactly the batch p = transform(u) it has just obtained;
    with paired=True transform() runs the one-launch cube -> theta -> log-L call and loglike() answers from it when handed
    that very batch (compared element by element; any other batch is evaluated as usual) — one round trip per proposal
    round instead of two (1024 points: 42 us instead of 34 + 34)."""
    if low_latency and not vectorized:
        model.scalar_server(True)
    if vectorized and paired:
        last = {"theta": None, "logl": None}

        def prior(hypercubes):
            theta, logl = model.prior_loglike_batch(np.asarray(hypercubes, dtype=np.float64))
            last["theta"], last["logl"] = theta.copy(), logl
            return theta

        def loglike(thetas):
            t = last["theta"]
            x = np.asarray(thetas, dtype=np.float64)
            if t is not None and x.shape == t.shape and np.array_equal(x, t):
                return last["logl"].copy()
            return model.log_likelihood_batch(x)
    elif vectorized:
        def prior(hypercubes):
            return model.prior_transform_batch(np.asarray(hypercubes, dtype=np.float64))

        def loglike(thetas):
            return model.log_likelihood_batch(np.asarray(thetas, dtype=np.float64))
    else:
        def prior(hypercube):
            return model.prior_transform(np.asarray(hypercube, dtype=np.float64))

        def loglike(x):
            return model.log_likelihood(x)

    return prior, loglike
