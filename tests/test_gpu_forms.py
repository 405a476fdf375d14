"""The two launch forms of the log-L kernel — 256-thread tiles and the CU-wide form (one 1024-thread workgroup per
CU, include/rvll.h rvll_set_kernel_form) — sum every point's contributions in the same order: results must be
bit-identical whatever form, chunking or share of the batch a workgroup gets.  Parity with the reference is then
inherited from the tile form's golden / oracle tests (tests/test_gpu_loglike.py), which run whichever form the
batch size selects."""
import numpy as np
import pytest

import golden
from evidence_amd import GpuRVModel
from evidence_amd.synthetic import make_workload

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(240)]


def _both(m, theta):
    out = {}
    for form in ("tile", "cu"):
        m.set_kernel_form(form)
        out[form] = m.log_likelihood_batch(theta, return_flags=True)
    m.set_kernel_form("auto")
    return out


@pytest.mark.parametrize("cfg,n", [(1, 400), (2, 4096), (3, 16384), (3, 16384 + 77), (3, 100), (3, 1), (4, 8192), (4, 3001), (5, 2048)])
def test_cu_form_is_bit_identical_to_tiles(gpu_required, cfg, n):
    w = make_workload(cfg)
    theta = w.sample_theta(n, seed=300 + cfg)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
        got = _both(m, theta)
        tm = None
        if n >= 4096:
            m.dev_upload_theta(theta)
            tm = m.dev_time_loglike(n, warmup=0, iters=1)
    assert np.array_equal(got["tile"][0], got["cu"][0])
    assert np.array_equal(got["tile"][1], got["cu"][1])
    assert np.isfinite(got["cu"][0]).all()
    if tm is not None and cfg >= 3 and n >= 8192:
        assert tm["threads"] == 1024          # large batches take the CU-wide form by default


@pytest.mark.parametrize("precision", ["mixed", "fp32"])
def test_cu_form_in_reduced_precision_same_values_to_float_rounding(gpu_required, precision):
    w = make_workload(3)
    theta = w.sample_theta(5000, seed=11)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, precision=precision) as m:
        got = _both(m, theta)
        again = _both(m, theta)
    # Round 4: the reduced-precision modes run their Newton iteration wave-wide (two items a lane, until every item of the wave
    # has met the stop rule: rvll_tile.h, eval_item_pair), so an item may take a step or two more than it needs, depending on
    # its neighbours in the wave: the same launch gives the same bits, another tiling the same values to a few float ulps of
    # the model.  (The parity mode, fp64, is bit-identical across forms: the tests around this one.)
    for form in ("tile", "cu"):
        assert np.array_equal(got[form][0], again[form][0])
    assert np.max(np.abs(got["tile"][0] - got["cu"][0]) / np.abs(got["tile"][0])) <= 3e-7


def test_cu_form_itmax_and_invalid_orbits(gpu_required):
    """The rare paths: a solve that hits itmax (nu stays 0 from that epoch on, trueanomaly.c:32-33) and invalid
    orbits (-1e30, rvmodel:203) go through the same redo / flag logic in both forms."""
    case = golden.config_case(3)
    theta = np.tile(case.theta, (40, 1))
    for itmax in (1, 2, 10000):
        with GpuRVModel(case.fixed, case.table, case.parnames, itmax=itmax) as m:
            got = _both(m, theta)
        assert np.array_equal(got["tile"][0], got["cu"][0]), itmax
        assert np.array_equal(got["tile"][1], got["cu"][1]), itmax
    inv = [c for c in golden.all_loglike_cases() if c.name.endswith("_invalid")]
    for c in inv:
        with GpuRVModel(c.fixed, c.table, c.parnames, linpar_dict=c.linpar or None) as m:
            got = _both(m, np.tile(c.theta, (8, 1)))
        assert np.array_equal(got["tile"][0], got["cu"][0]) and np.all(got["cu"][0] == -1e30)


@pytest.mark.parametrize("n_epochs", [1000, 4096, 4500, 9000])
def test_cu_form_across_the_tile_window(gpu_required, n_epochs):
    """Beyond 4096 epochs the tile form sums a point window by window (slices at multiples of 4096); the CU-wide
    form keeps the whole point in LDS and must reproduce that order."""
    from test_gpu_loglike import _synthetic_case
    rng = np.random.default_rng(n_epochs)
    table, free, fixed, ranges, linpar = _synthetic_case(rng, n_epochs, 2, 2, False, 0, False)
    theta = np.stack([rng.uniform(*ranges[nm], 301) for nm in free], axis=1)
    with GpuRVModel(fixed, table, free) as m:
        got = _both(m, theta)
    assert np.array_equal(got["tile"][0], got["cu"][0])


def test_every_golden_case_in_cu_form(gpu_required):
    """All parametrisations, drift orders, linear terms and instruments of the golden set, forced through the
    CU-wide form, against the reference's own numbers."""
    for case in golden.all_loglike_cases():
        with GpuRVModel(case.fixed, case.table, case.parnames, linpar_dict=case.linpar or None) as m:
            m.set_kernel_form("cu")
            got = m.log_likelihood_batch(case.theta)
        err = golden.rel_err(got, case.logL)
        assert err.max() <= 1e-10, (case.name, float(err.max()))


@pytest.mark.parametrize("n_epochs,nplanets,ninst,drift,nlin,only_offset,npts", [
    (1, 1, 1, False, 0, False, 300),          # a single epoch: 64 points per wave round
    (3, 0, 2, True, 0, False, 513),           # no planet at all, drift only
    (17, 2, 1, False, 1, False, 1000),        # several points per wave round
    (63, 8, 5, True, 3, False, 257),          # many planets / instruments / linear terms, ragged batch
    (64, 1, 1, False, 0, False, 129), (65, 1, 3, True, 2, False, 129),
    (127, 3, 2, False, 0, False, 2049),
    (200, 2, 2, False, 0, True, 1000),        # exactly one free parameter
    (2500, 4, 3, True, 1, False, 700),        # big tiles: a few points fill the LDS budget
    (15000, 1, 1, False, 0, False, 40),       # one point per tile, four 4096-slices per point
])
def test_unusual_model_shapes_in_both_forms(gpu_required, n_epochs, nplanets, ninst, drift, nlin, only_offset, npts):
    """Random model shapes through the tile form, the CU-wide form (forced) and the oracle: the wave-round index
    arithmetic (several points per round when there are fewer than 64 epochs), tile sizing against the LDS budget
    and the per-point slices all have to agree."""
    from oracle.oracle import OracleModel
    from test_gpu_loglike import _synthetic_case
    rng = np.random.default_rng(n_epochs * 37 + nplanets)
    table, free, fixed, ranges, linpar = _synthetic_case(rng, n_epochs, nplanets, ninst, drift, nlin, only_offset)
    theta = np.stack([rng.uniform(*ranges[nm], npts) for nm in free], axis=1)
    with GpuRVModel(fixed, table, free, linpar_dict=linpar or None) as m:
        got = _both(m, theta)
        layout = m.layout
        m.set_kernel_form("cu")
        m.dev_upload_theta(theta)
        tm = m.dev_time_loglike(npts, warmup=0, iters=1)
    assert np.array_equal(got["tile"][0], got["cu"][0]) and np.array_equal(got["tile"][1], got["cu"][1])
    assert tm["threads"] == 1024                      # every one of these shapes fits the CU-wide form
    series = np.stack([linpar[k] for k in layout.linpar_names]) if layout.linpar_names else None
    ref = OracleModel(layout, table, series).loglike(theta, nthreads=8)
    assert golden.rel_err(got["cu"][0], ref).max() <= 1e-10
