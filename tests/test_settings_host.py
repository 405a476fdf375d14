"""CPU: sampler settings as configuration — mirrors the reference's own test of set_polysettings
(tests/test_polychord.py:33-73: 51 Peg, 7 free parameters -> nlive 175, num_repeats 35, wrong types -> TypeError) and
the defaults of set_ultrasettings (evidence/ultranest/__init__.py:333-338)."""
import numpy as np
import pytest

from evidence_amd.settings import polychord_defaults, ultranest_defaults


def test_polysettings_defaults_and_type_checks_as_the_reference_tests_them():
    ndim = 7                                         # config_51Peg_example.py with one planet: 5 + offset + jitter
    settings = polychord_defaults(ndim, None)
    assert settings["nlive"] == 175
    assert settings["num_repeats"] == 35
    assert settings["do_clustering"] is True
    assert settings["precision_criterion"] == 0.001
    assert settings["write_resume"] is False and settings["read_resume"] is False
    assert settings["feedback"] == 1 and settings["boost_posterior"] == 0.0
    for bad in ({"nlive": 43.7}, {"num_repeats": ["5"]}, {"do_clustering": {"clustering": True}},
                {"read_resume": 74}, {"precision_criterion": False}, {"nlive": True}):
        with pytest.raises(TypeError):
            polychord_defaults(ndim, bad)
    with pytest.raises(TypeError):
        polychord_defaults(ndim, [("nlive", 100)])
    assert polychord_defaults(ndim, {"nlive": 100, "feedback": 0})["nlive"] == 100
    assert polychord_defaults(ndim, {"anything_else": "is passed through"})["anything_else"] == "is passed through"


def test_ultrasettings_defaults():
    s = ultranest_defaults(19)
    assert s == {"nlive": 475, "nsteps": 57, "dlogz": 0.5, "frac_remain": 0.01, "num_bootstraps": 30}
    assert ultranest_defaults(19, {"nlive": 50})["nlive"] == 50
    with pytest.raises(TypeError):
        ultranest_defaults(19, "nlive=50")


def test_the_in_repo_driver_starts_from_the_reference_defaults():
    """nested.run_nested_slice without nlive / nsteps uses 25 ndim live points and 3 ndim moves per new point."""
    from evidence_amd.nested import run_nested_slice
    seen = {}

    def prior(u):
        seen.setdefault("nlive", len(u))
        return 20.0 * u - 10.0

    def loglike(t):
        return -0.5 * (t ** 2).sum(axis=1) - np.log(2 * np.pi)

    def walker(cube, theta, logl, lstar, chol, wrapped, nsteps, max_rounds, seed):
        seen.setdefault("nsteps", nsteps)
        raise StopIteration

    with pytest.raises(StopIteration):
        run_nested_slice(prior, loglike, 2, walker=walker)
    assert seen == {"nlive": 50, "nsteps": 6}


def test_the_samplers_sort_is_the_stable_sort_whatever_route_it_takes():
    """nested._stable_argsort goes through the vectorised unstable sort only when that provably gives the stable
    permutation (no equal neighbours in sorted order, no NaN)."""
    import numpy as np
    from evidence_amd.nested import _stable_argsort
    rng = np.random.default_rng(3)
    x = rng.normal(size=5000)
    assert np.array_equal(_stable_argsort(x), np.argsort(x, kind="stable"))
    x[[5, 7, 4000]] = -1e30                      # ties (several invalid orbits): the stable order of equal keys matters
    x[100] = x[200]
    assert np.array_equal(_stable_argsort(x), np.argsort(x, kind="stable"))
    x[9] = np.nan
    x[4999] = np.nan
    assert np.array_equal(_stable_argsort(x), np.argsort(x, kind="stable"))
    assert _stable_argsort(np.empty(0)).size == 0
