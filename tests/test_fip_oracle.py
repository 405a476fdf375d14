"""The FIP oracle against periodograms written by the reference's own script (bit-exact)."""
import numpy as np
import pytest

from oracle import fip_oracle
from tests.fip_cases import CASES, FipCase


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_reference_periodogram(name):
    c = FipCase(name)
    nu, nua, nub = fip_oracle.frequency_grid(c.pmin, c.pmax, c.tobs)
    assert np.array_equal(nu, c.nu)
    pky = fip_oracle.model_probabilities(c.logzs)
    assert abs(pky.sum() - 1.0) < 1e-12
    fapnu = fip_oracle.accumulate(c.posteriors, pky, nua, nub)
    assert fapnu.shape == c.fapnu.shape
    assert np.array_equal(fapnu, c.fapnu)


@pytest.mark.parametrize("name", CASES)
def test_c_fold_reproduces_reference_periodogram(name):
    """rvo_fip_accumulate (the fold used for large inputs and as the CPU baseline) on the rows the host
    flattens — against the reference's file, bit for bit."""
    from evidence_amd import fip
    from oracle import oracle
    c = FipCase(name)
    _, nua, nub = fip.frequency_grid(c.pmin, c.pmax, c.tobs)
    pky = fip.model_probabilities(c.logzs)
    periods, contrib, run_start = fip.flatten_posteriors(c.posteriors, pky)
    assert run_start[0] == 0 and run_start[-1] == len(contrib) == len(periods)
    assert np.array_equal(oracle.fip_accumulate(nua, nub, periods, contrib, run_start), c.fapnu)


def test_host_helpers_match_oracle_restatement():
    from evidence_amd import fip
    c = FipCase("edges")
    for a, b in zip(fip.frequency_grid(c.pmin, c.pmax, c.tobs), fip_oracle.frequency_grid(c.pmin, c.pmax, c.tobs)):
        assert np.array_equal(a, b)
    assert np.array_equal(fip.model_probabilities(c.logzs), fip_oracle.model_probabilities(c.logzs))
    import pandas as pd
    dd = {"a": {"data": pd.DataFrame({"rjd": c.times[::2]})}, "b": {"data": pd.DataFrame({"jdb": c.times[1::2]})}}
    assert fip.observation_span(dd) == c.tobs
    s = fip.fip_summary(c.fapnu, c.nu)
    o = fip_oracle.summary(c.fapnu)
    for k in ("log10fips", "diffs", "median", "std", "mean"):
        assert np.array_equal(s[k], o[k])
    assert s["converged"] == bool((o["diffs"] <= 1).all())


def test_flatten_layout_and_padding():
    from evidence_amd import fip
    rng = np.random.default_rng(0)
    post = [[None, (rng.uniform(1, 9, (5, 1)), rng.random(5)), (rng.uniform(1, 9, (7, 2)), rng.random(7))],
            [None, (rng.uniform(1, 9, (4, 1)), rng.random(4)), (rng.uniform(1, 9, (3, 2)), rng.random(3))]]
    pky = np.array([0.1, 0.3, 0.6])
    periods, contrib, run_start = fip.flatten_posteriors(post, pky)
    assert periods.shape == (19, 2) and list(run_start) == [0, 12, 19]
    assert np.isnan(periods[:5, 1]).all() and not np.isnan(periods[5:12]).any()
    w = post[0][2][1]
    assert np.array_equal(contrib[5:12], pky[2] * (w / np.sum(w)))
    assert np.array_equal(post[0][2][1], w)                      # caller's weights are not normalised in place
    with pytest.raises(ValueError):
        fip.flatten_posteriors([[None, (np.ones((3, 9)), np.ones(3))]], np.array([0.5, 0.5]))
