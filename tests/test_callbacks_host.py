"""CPU: the sampler-facing closures (evidence_amd/callbacks.py) over a stand-in model — return conventions of the
reference's wrappers (evidence/polychord/__init__.py:130-171, evidence/ultranest/__init__.py:125-146) and the rule of the
paired forms: loglike() answers from the pair only when it is handed exactly what prior() returned."""
import numpy as np

from evidence_amd.callbacks import make_polychord_callbacks, make_ultranest_callbacks, wrapped_params


class StandIn:
    """theta = 10 * cube, log-L = -sum(theta^2); counts the calls of each entry point."""
    parnames = ["a_offset", "planet1_ml0", "planet1_omega", "planet1_period"]
    ndim = 4

    def __init__(self):
        self.calls = {"prior": 0, "loglike": 0, "pair": 0, "server": None}

    def scalar_server(self, on):
        self.calls["server"] = bool(on)

    def prior_transform(self, cube):
        self.calls["prior"] += 1
        return 10.0 * np.asarray(cube)

    def log_likelihood(self, x):
        self.calls["loglike"] += 1
        return float(-np.sum(np.asarray(x) ** 2))

    def prior_loglike(self, cube):
        self.calls["pair"] += 1
        th = 10.0 * np.asarray(cube)
        return th, float(-np.sum(th ** 2))

    def prior_transform_batch(self, cubes):
        self.calls["prior"] += 1
        return 10.0 * np.asarray(cubes)

    def log_likelihood_batch(self, X):
        self.calls["loglike"] += 1
        return -np.sum(np.asarray(X) ** 2, axis=1)

    def prior_loglike_batch(self, cubes):
        self.calls["pair"] += 1
        th = 10.0 * np.asarray(cubes)
        return th, -np.sum(th ** 2, axis=1)


def test_polychord_conventions_and_the_paired_low_latency_form():
    m = StandIn()
    prior, loglike, ndim, nderived = make_polychord_callbacks(m)
    assert (ndim, nderived) == (4, 0) and m.calls["server"] is None
    th = prior(np.full(4, 0.1))
    assert np.allclose(th, 1.0) and loglike(th) == (-4.0, [])               # (logL, derived) as PolyChord wants it
    assert m.calls == {"prior": 1, "loglike": 1, "pair": 0, "server": None}

    m = StandIn()
    prior, loglike, _, _ = make_polychord_callbacks(m, low_latency=True)
    assert m.calls["server"] is True
    th = prior(np.full(4, 0.2))
    assert loglike(th) == (-16.0, []) and m.calls["pair"] == 1 and m.calls["loglike"] == 0     # answered from the pair
    assert loglike(th.tolist())[0] == -16.0 and m.calls["loglike"] == 0                       # the same values, any container
    other = th.copy(); other[2] += 1e-12
    assert loglike(other)[0] != -16.0 and m.calls["loglike"] == 1                             # anything else: evaluated
    th[0] = 99.0                                                                              # the caller's array is its own
    assert loglike(np.full(4, 2.0))[0] == -16.0 and m.calls["loglike"] == 1
    assert loglike(np.full(3, 2.0))[0] == -12.0 and m.calls["loglike"] == 2                   # another shape


def test_ultranest_conventions_and_the_paired_vectorized_form():
    m = StandIn()
    prior, loglike = make_ultranest_callbacks(m)
    assert isinstance(loglike(prior(np.full(4, 0.1))), float)                                 # a plain float
    vprior, vloglike = make_ultranest_callbacks(m, vectorized=True)
    cubes = np.random.default_rng(0).random((7, 4))
    assert vloglike(vprior(cubes)).shape == (7,)

    m = StandIn()
    vprior, vloglike = make_ultranest_callbacks(m, vectorized=True, paired=True)
    th = vprior(cubes)
    ll = vloglike(th)
    assert m.calls == {"prior": 0, "loglike": 0, "pair": 1, "server": None}
    assert np.array_equal(ll, -np.sum(th ** 2, axis=1))
    ll[0] = 0.0                                                                               # handed-out results are copies
    assert vloglike(th)[0] != 0.0 and m.calls["loglike"] == 0
    assert vloglike(th[:3]).shape == (3,) and m.calls["loglike"] == 1                         # another batch: evaluated
    th2 = th.copy(); th2[5, 1] = np.nan
    vloglike(th2)
    assert m.calls["loglike"] == 2                                                            # NaN never matches


def test_wrapped_parameters_are_the_angles_the_reference_wraps():
    # evidence/ultranest/__init__.py:159-163: 'omega' or 'ml0' in the name
    assert wrapped_params(StandIn.parnames).tolist() == [False, True, True, False]
