"""GPU parity of rvll_fip_accumulate: bit-identical to the periodograms the reference script wrote, and to
the C fold on larger synthetic posteriors (overlapping windows, NaN padding, empty runs, ragged tiles)."""
import numpy as np
import pytest

from evidence_amd import fip
from tests.fip_cases import CASES, FipCase

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(240)]


@pytest.mark.parametrize("name", CASES)
def test_reference_periodogram_bit_exact(gpu_required, name):
    c = FipCase(name)
    nu, nua, nub = fip.frequency_grid(c.pmin, c.pmax, c.tobs)
    assert np.array_equal(nu, c.nu)
    pky = fip.model_probabilities(c.logzs)
    got = fip.fip_periodogram(c.posteriors, pky, nua, nub)
    assert got.shape == c.fapnu.shape
    assert np.array_equal(got, c.fapnu), float(np.abs(got - c.fapnu).max())


def _synthetic(rng, runs, nmod, n, pmin, pmax, nan_rows=False):
    post = []
    for _ in range(runs):
        per_k = [None]
        for k in range(1, nmod):
            m = int(n * (0.6 + 0.8 * rng.random()))
            s = np.exp(rng.uniform(np.log(pmin * 0.8), np.log(pmax * 1.2), (m, k)))
            peak = np.exp(rng.uniform(np.log(pmin), np.log(pmax), k))
            pick = rng.random((m, k)) < 0.6
            s = np.where(pick, peak * np.exp(rng.normal(0, 1e-3, (m, k))), s)
            if k > 1:
                s[: m // 5, 1] = s[: m // 5, 0] * (1 + rng.normal(0, 3e-5, m // 5))     # overlapping windows
            if nan_rows:
                s[rng.integers(0, m, 5), 0] = np.nan
                s[rng.integers(0, m, 3), 0] = 0.0
                s[rng.integers(0, m, 3), 0] = -3.0
                s[rng.integers(0, m, 2), 0] = np.inf
            per_k.append((s, rng.gamma(0.4, 1.0, m)))
        post.append(per_k)
    return post


@pytest.mark.parametrize("runs, nmod, n, nfreq, tobs", [
    (4, 4, 30000, 50000, 900.0),          # the script's grid, realistic posterior sizes
    (1, 2, 70000, 50000, 3000.0),         # one run, one planet, narrow windows
    (3, 6, 4000, 12345, 40.0),            # wide windows (hundreds of bins), nfreq not a multiple of the tile
    (2, 9, 1500, 777, 200.0),             # eight periods per sample (the ABI maximum)
])
def test_large_posteriors_match_c_fold(gpu_required, runs, nmod, n, nfreq, tobs):
    from oracle import oracle
    rng = np.random.default_rng(nmod * 1000 + runs)
    pmin, pmax = 1.3, 400.0
    _, nua, nub = fip.frequency_grid(pmin, pmax, tobs, nfreq=nfreq)
    post = _synthetic(rng, runs, nmod, n, pmin, pmax, nan_rows=True)
    pky = rng.dirichlet(np.ones(nmod))
    got = fip.fip_periodogram(post, pky, nua, nub)
    periods, contrib, run_start = fip.flatten_posteriors(post, pky)
    want = oracle.fip_accumulate(nua, nub, periods, contrib, run_start)
    assert np.array_equal(got, want), float(np.abs(got - want).max())
    assert (got <= 1.0).all() and got.min() < 1.0


def test_degenerate_inputs(gpu_required):
    _, nua, nub = fip.frequency_grid(2.0, 100.0, 500.0, nfreq=1000)
    # a run without any posterior rows keeps its ones; a run whose samples all fall outside the grid too
    post = [[None], [None, (np.array([[1.0], [1000.0], [np.nan]]), np.ones(3))]]
    got = fip.fip_periodogram(post, np.array([0.5, 0.5]), nua, nub)
    assert got.shape == (2, 1000) and (got == 1.0).all()
    # repeats only re-run the kernels from the same input
    post = _synthetic(np.random.default_rng(5), 2, 3, 2000, 2.0, 100.0)
    a = fip.fip_periodogram(post, np.array([0.2, 0.3, 0.5]), nua, nub)
    b, t = fip.fip_periodogram(post, np.array([0.2, 0.3, 0.5]), nua, nub, repeats=3, return_timing=True)
    assert np.array_equal(a, b) and t["repeats"] == 3 and t["accumulate_ms"] > 0
    from evidence_amd import RvllError
    with pytest.raises(RvllError):
        fip.fip_periodogram(post, np.array([0.2, 0.3, 0.5]), nua[::-1].copy(), nub)
