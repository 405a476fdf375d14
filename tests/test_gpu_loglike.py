"""GPU parity: the HIP log-L path (through the C-ABI) against
  (a) the golden vectors the reference itself produced      -> <= 1e-10 relative (north star)
  (b) the CPU oracle on larger seeded batches               -> <= 1e-10 relative, flips located
  (c) size-independent properties at BASELINE.json batch sizes.
"""
import numpy as np
import pytest

import golden
from evidence_amd import GpuRVModel, FLAG_INVALID_ORBIT, FLAG_WANDERED
from evidence_amd.synthetic import make_workload

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(240)]
TOL = 1e-10     # BASELINE.json north_star: <= 1e-10 relative on identical theta

CASES = golden.all_loglike_cases()


def _model(case, **kw):
    return GpuRVModel(case.fixed, case.table, case.parnames, linpar_dict=case.linpar or None, **kw)


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_hip_matches_reference_golden(gpu_required, case):
    with _model(case) as m:
        got = m.log_likelihood_batch(case.theta)
    err = golden.rel_err(got, case.logL)
    assert err.max() <= TOL, (case.name, float(err.max()), int(err.argmax()))


def test_known_answer_51peg(gpu_required):
    case = golden.peg51_cases()[0]
    with _model(case) as m:
        got = m.log_likelihood(case.theta[0])
    assert abs(got - (-11539.57252446112)) <= TOL * 11539.6


def test_invalid_orbit_is_minus_1e30_with_flag(gpu_required):
    for case in CASES:
        if case.name.endswith("_invalid"):
            with _model(case) as m:
                got, flags = m.log_likelihood_batch(case.theta, return_flags=True)
            assert np.all(got == -1e30) and np.all(flags & FLAG_INVALID_ORBIT)


@pytest.mark.parametrize("cfg,n", [(1, 400), (2, 4096), (3, 16384), (4, 2048), (5, 512)])
def test_hip_matches_oracle_on_seeded_batches(gpu_required, cfg, n):
    from oracle.oracle import OracleModel
    w = make_workload(cfg)
    theta = w.sample_theta(n, seed=90 + cfg)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
        got, flags = m.log_likelihood_batch(theta, return_flags=True)
        layout = m.layout
    om = OracleModel(layout, w.table)
    ref, rflags = om.loglike(theta, nthreads=8, return_flags=True)
    err = golden.rel_err(got, ref)
    bad = np.flatnonzero(err > TOL)
    detail = ""
    if bad.size:                         # locate Newton step-count flips (SURVEY.md §7 hard part 1)
        i = int(bad[0])
        detail = f"point {i}: err {err[i]:.3e}, steps max {om.iteration_counts(theta[i]).max()}"
    assert bad.size == 0, (cfg, bad.size, detail)
    assert np.percentile(err, 99.9) <= 1e-12
    # flags: nothing invalid, nothing at itmax; RVLL_FLAG_WANDERED exactly where the oracle counted a solve of > 8 steps
    assert np.array_equal(flags, rflags) and not (flags & ~FLAG_WANDERED).any()


@pytest.mark.parametrize("pb", [1, 2, 3, 5, 8, 16, 20, 32])
def test_result_independent_of_launch_geometry(gpu_required, pb):
    """points-per-workgroup only changes the tiling; every log-L is reduced in the same order."""
    w = make_workload(3)
    theta = w.sample_theta(777, seed=5)          # ragged: not a multiple of any pb
    with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
        base = m.log_likelihood_batch(theta)
        m.set_points_per_block(pb)
        got = m.log_likelihood_batch(theta)
    assert np.array_equal(got, base) or golden.rel_err(got, base).max() <= 1e-14


@pytest.mark.parametrize("n_epochs, pbs", [(1000, (1, 3, 4, 5, 9)), (2000, (1, 2, 3, 5)), (4500, (1, 2, 3))])
def test_result_independent_of_geometry_across_lds_windows(gpu_required, n_epochs, pbs):
    """A tile larger than the LDS window is summed window by window; the windows are cut at point-local
    positions, so the bits do not depend on how many points share a workgroup (nor on the shard size that
    picked that number on a multi-GPU run)."""
    rng = np.random.default_rng(n_epochs)
    table, free, fixed, ranges, linpar = _synthetic_case(rng, n_epochs, 2, 2, False, 0, False)
    theta = np.stack([rng.uniform(*ranges[nm], 301) for nm in free], axis=1)
    with GpuRVModel(fixed, table, free) as m:
        base = m.log_likelihood_batch(theta)
        for pb in pbs:
            m.set_points_per_block(pb)
            assert np.array_equal(m.log_likelihood_batch(theta), base), pb


def test_batch_of_one_and_scalar_callback(gpu_required):
    case = golden.config_case(3)
    with _model(case) as m:
        full = m.log_likelihood_batch(case.theta)
        singles = np.array([m.log_likelihood(x) for x in case.theta[:16]])
        empty = m.log_likelihood_batch(np.empty((0, m.ndim)))
    assert empty.shape == (0,)
    assert golden.rel_err(singles, full[:16]).max() <= 1e-14


def test_deterministic_across_calls(gpu_required):
    w = make_workload(3)
    theta = w.sample_theta(5000, seed=1)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
        a = m.log_likelihood_batch(theta)
        b = m.log_likelihood_batch(theta)
    assert np.array_equal(a, b)


def test_permutation_equivariance_at_full_batch(gpu_required):
    """Size-independent property at the BASELINE.json batch size: permuting live points permutes
    log-L; duplicated points give identical log-L."""
    w = make_workload(3)
    n = w.batch                                   # 16384
    theta = w.sample_theta(n, seed=2)
    perm = np.random.default_rng(0).permutation(n)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
        a = m.log_likelihood_batch(theta)
        b = m.log_likelihood_batch(theta[perm])
    assert np.array_equal(a[perm], b)


@pytest.mark.parametrize("cfg", [4, 5])
def test_shard_invariance_at_full_multi_gpu_batch(gpu_required, cfg):
    """BASELINE.json configs[3]/[4] at their full size (65 536 / 131 072 live points): evaluating the whole
    batch in one call equals evaluating the 8 contiguous shards the multi-GPU step hands to its ranks, bit
    for bit (each shard picks its own launch geometry), and a sample of rows matches the oracle."""
    from evidence_amd.sharded import partition
    from oracle.oracle import OracleModel
    w = make_workload(cfg)
    n = w.batch
    theta = w.sample_theta(n, seed=40 + cfg)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
        whole = m.log_likelihood_batch(theta)
        parts = np.concatenate([m.log_likelihood_batch(theta[lo:hi]) for lo, hi in partition(n, 8)])
        odd = np.concatenate([m.log_likelihood_batch(theta[lo:hi]) for lo, hi in partition(n, 7)])
        layout = m.layout
    assert whole.shape == (n,) and np.isfinite(whole).all()
    assert np.array_equal(whole, parts) and np.array_equal(whole, odd)
    rows = np.random.default_rng(cfg).choice(n, 96, replace=False)
    ref = OracleModel(layout, w.table).loglike(theta[rows], nthreads=8)
    assert golden.rel_err(whole[rows], ref).max() <= TOL


def test_offset_shift_property(gpu_required):
    """Shifting every vrad and every instrument offset by the same constant leaves log-L unchanged
    (to rounding): a property that needs no reference values."""
    w = make_workload(3)
    theta = w.sample_theta(2048, seed=3)
    shift = 3.25
    from evidence_amd.data import EpochTable
    t2 = EpochTable.from_arrays(w.table.insts, w.table.time, w.table.vrad + shift, w.table.svrad, w.table.inst_id)
    th2 = theta.copy()
    for i, nme in enumerate(w.parnames):
        if nme.endswith("_offset"):
            th2[:, i] += shift
    with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m1, GpuRVModel(w.fixedpardict, t2, w.parnames) as m2:
        a, b = m1.log_likelihood_batch(theta), m2.log_likelihood_batch(th2)
    assert golden.rel_err(a, b).max() <= 1e-11


def test_itmax_abort_semantics_match_oracle(gpu_required):
    """trueanomaly.c:32-33: when a solve reaches itmax the planet's array is abandoned there and nu stays 0
    from that epoch on.  Forced with a tiny itmax; the kernel must agree with the oracle point by point."""
    from oracle.oracle import OracleModel
    case = golden.config_case(3)
    for itmax in (1, 2, 3):
        with _model(case, itmax=itmax) as m:
            got, flags = m.log_likelihood_batch(case.theta, return_flags=True)
            layout = m.layout
        ref, rflags = OracleModel(layout, case.table).loglike(case.theta, return_flags=True)
        assert np.array_equal(flags & 2, rflags & 2), itmax
        assert (flags & 2).any()
        assert golden.rel_err(got, ref).max() <= TOL, (itmax, float(golden.rel_err(got, ref).max()))


def test_million_point_batch_and_nonfinite_inputs(gpu_required):
    """A batch far beyond the BASELINE sizes (10^6 live points, 160 MB of theta): grid sizing, buffer growth
    and shrink-back; a strided sample is checked against the oracle.  Non-finite theta rows must come back
    non-finite (NaN) without disturbing their neighbours."""
    from oracle.oracle import OracleModel
    w = make_workload(3)
    n = 1_000_000
    theta = np.tile(w.sample_theta(50_000, seed=8), (20, 1))
    theta[12345, 3] = np.nan
    theta[777_777, 5] = np.inf
    with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
        got = m.log_likelihood_batch(theta)
        small = m.log_likelihood_batch(theta[:10])           # capacity stays, smaller batch still right
        layout = m.layout
    assert got.shape == (n,) and np.array_equal(small, got[:10])
    assert np.isnan(got[12345]) and not np.isfinite(got[777_777])
    assert np.isfinite(got[12344]) and np.isfinite(got[12346])
    idx = np.arange(0, n, 997)
    idx = idx[(idx != 12345) & (idx != 777_777)]
    ref = OracleModel(layout, w.table).loglike(theta[idx], nthreads=8)
    assert golden.rel_err(got[idx], ref).max() <= TOL
    assert np.array_equal(got[100_000:150_000], got[150_000:200_000])    # clean tiled copies agree bit for bit


def test_no_step_count_flips_in_a_quarter_billion_solves(gpu_required):
    """the kernel replicates the reference's (unconverged, tol = 1e-4) Newton rule so closely that
    a step-count flip (which moves log-L by ~1e-9 relative) is a ~1e-12-per-solve event.  400 000 live points
    x 3 planets x 200 epochs = 2.4e8 solves against the oracle: every point is within the 1e-10 bar; every
    point whose eccentricities are all <= 0.95 agrees to 1e-13 (no flip anywhere).  Beyond e ~ 0.97 Newton
    from E = M wanders chaotically for tens to hundreds of steps (up to 1159) and amplifies
    1-ulp differences, so the stop can land a step apart there: observed worst case 5.8e-12."""
    from oracle.oracle import OracleModel
    w = make_workload(3)
    n = 400_000
    theta = w.sample_theta(n, seed=2024)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
        got = m.log_likelihood_batch(theta)
        layout = m.layout
    ref = OracleModel(layout, w.table).loglike(theta, nthreads=16)
    err = golden.rel_err(got, ref)
    assert err.max() <= TOL, (float(err.max()), int(err.argmax()))
    ecc = theta[:, [w.parnames.index(f"planet{k}_ecc") for k in (1, 2, 3)]].max(axis=1)
    calm = ecc <= 0.95
    assert calm.mean() > 0.97
    assert err[calm].max() <= 1e-13, float(err[calm].max())
    assert np.percentile(err, 99.99) <= 1e-14


def _synthetic_case(rng, n_epochs, nplanets, ninst, drift=False, nlin=0, nfree_only_offset=False):
    """A model of arbitrary shape with every planet/instrument parameter free (or only one offset free)."""
    from evidence_amd.data import EpochTable
    t = np.sort(rng.uniform(50000.0, 53000.0, n_epochs))
    inst = rng.integers(0, ninst, n_epochs)
    inst[:ninst] = np.arange(ninst)
    order = np.argsort(inst, kind="stable")
    names_i = [f"i{k}" for k in range(ninst)]
    table = EpochTable.from_arrays(names_i, t[order], rng.normal(0, 10, n_epochs), rng.uniform(0.5, 3, n_epochs),
                                   inst[order].astype(np.int32))
    free, fixed, ranges = [], {}, {}
    for p in range(1, nplanets + 1):
        for suf, lo, hi in (("k1", 0.5, 30), ("period", 2, 300), ("ecc", 0, 0.8), ("omega", 0, 6.28), ("ma0", 0, 6.28)):
            free.append(f"planet{p}_{suf}"); ranges[free[-1]] = (lo, hi)
        fixed[f"planet{p}_epoch"] = 51000.0 + p
    for nm in names_i:
        free += [f"{nm}_offset", f"{nm}_jitter"]
        ranges[f"{nm}_offset"], ranges[f"{nm}_jitter"] = (-5, 5), (0, 5)
    if drift:
        free += ["drift_lin", "drift_quad"]; ranges["drift_lin"] = ranges["drift_quad"] = (-2, 2)
        fixed["drift_tref"] = 51500.0
    linpar = {}
    for k in range(nlin):
        free.append(f"linpar_s{k}"); ranges[free[-1]] = (-2, 2)
        linpar[f"s{k}"] = rng.normal(0, 1, n_epochs)
    if nfree_only_offset:
        keep = f"{names_i[0]}_offset"
        for nm in free:
            if nm != keep:
                fixed[nm] = 0.5 * sum(ranges[nm])
        free = [keep]
    return table, sorted(free), fixed, ranges, linpar


@pytest.mark.parametrize("n_epochs,nplanets,ninst,drift,nlin,only_offset,npts", [
    (10_000, 1, 2, False, 0, False, 96),      # one live point spans three 4096-slot LDS windows
    (5_000, 2, 1, True, 0, False, 64),        # ... with drift, two planets
    (1, 1, 1, False, 0, False, 300),          # a single epoch
    (63, 8, 5, True, 3, False, 257),          # many planets / instruments / linear terms, ragged batch
    (200, 2, 2, False, 0, True, 1000),        # exactly one free parameter, everything else fixed
])
def test_unusual_model_shapes_match_oracle(gpu_required, n_epochs, nplanets, ninst, drift, nlin, only_offset, npts):
    from oracle.oracle import OracleModel
    rng = np.random.default_rng(n_epochs * 31 + nplanets)
    table, free, fixed, ranges, linpar = _synthetic_case(rng, n_epochs, nplanets, ninst, drift, nlin, only_offset)
    theta = np.stack([rng.uniform(*ranges[nm], npts) for nm in free], axis=1)
    with GpuRVModel(fixed, table, free, linpar_dict=linpar or None) as m:
        got = m.log_likelihood_batch(theta)
        layout = m.layout
        for pb in (1, 2, 7):
            m.set_points_per_block(pb)
            assert golden.rel_err(m.log_likelihood_batch(theta), got).max() <= 1e-13
    series = np.stack([linpar[k] for k in layout.linpar_names]) if layout.linpar_names else None
    ref = OracleModel(layout, table, series).loglike(theta, nthreads=8)
    assert golden.rel_err(got, ref).max() <= TOL, float(golden.rel_err(got, ref).max())


def test_high_eccentricity_parity_is_the_references_own_conditioning(gpu_required):
    """Where Newton from E = M is least forgiving (one planet at e = 0.95 .. 0.9925; at the 0.99 clamp the iteration is
    thrown out to |E| ~ 1e9 .. 1e22 and finds its way back): the device follows the reference's path — every
    sin / cos out there is reduced exactly and, for a solve that wanders, rounded correctly (rvll_math.h, sincos_cr) — far
    closer than the reference's own value is worth: the oracle with its sin / cos nudged by one unit in the last place moves
    by up to ~1e-9 on 1-3 % of such points; the device is beyond 1e-10 on ~0.01 %.  Points none of whose solves wanders
    (<= 12 steps) meet 1e-10."""
    from oracle.oracle import OracleModel
    case = golden.high_ecc_case()
    names = case.parnames
    rng = np.random.default_rng(7)
    n = 6000
    theta = rng.uniform(case.theta.min(axis=0), case.theta.max(axis=0), (n, len(names)))
    ie = names.index("planet1_ecc")
    theta[:, ie] = rng.uniform(0.95, 0.9925, n)
    theta[:, names.index("planet2_ecc")] = rng.beta(0.867, 3.03, n)
    with GpuRVModel(case.fixed, case.table, names) as m:
        got, flags = m.log_likelihood_batch(theta, return_flags=True)
        layout = m.layout
    om = OracleModel(layout, case.table)
    ref, rflags = om.loglike(theta, nthreads=8, return_flags=True)
    cond = np.maximum(om.conditioning(theta, nthreads=8, eps=-2.0 ** -53), om.conditioning(theta, nthreads=8, eps=2.0 ** -52))
    err = golden.rel_err(got, ref)
    # Round 4: a solve that wanders is done again, from its start, with CORRECTLY ROUNDED sin / cos (rvll_math.h, sincos_cr; the
    # redo pass of the tile) — what glibc's are nearly always.  Measured on 20000 such points (profiles/r04_high_ecc_parity.txt):
    # beyond 1e-10: 90 -> 2, the largest 6.6e-10 -> 3.3e-10.  (Round 3's bounds here: 5e-9, and "no more often than twice the
    # reference from itself one ulp of libm away".)
    assert err.max() <= 1e-9, float(err.max())                       # (this sample: one point, 7.0e-10; round 3: 5e-9 allowed)
    assert (err > TOL).sum() <= max(2, (cond > TOL).sum() // 10)      # an order of magnitude rarer than the reference from itself
    # wherever no solve of the point wanders (the oracle's own step counts: <= 12 everywhere), the plain bar holds
    worst = np.argsort(-err)[:200]
    for i in worst:
        if err[i] > TOL:
            assert om.iteration_counts(theta[i]).max() > 12, (int(i), float(err[i]))
    calm = theta[:, ie] < 0.965
    assert err[calm].max() <= 1e-11
    # The contract of RVLL_FLAG_WANDERED (include/rvll.h): a caller can tell which values are in that
    # class.  The bit is set exactly where the oracle's own step counts exceed 8; EVERY point without it meets the plain
    # bar, and every point beyond the bar carries it.
    wandered = (flags & FLAG_WANDERED) != 0
    assert np.array_equal(flags, rflags)
    assert err[~wandered].max() <= TOL, float(err[~wandered].max())
    assert wandered[err > TOL].all()
    assert 0 < wandered.sum() < n                                    # (both classes are present in this sample)
    for i in np.flatnonzero(wandered)[:40]:
        assert om.iteration_counts(theta[i]).max() > 8
    for i in np.flatnonzero(~wandered)[:40]:
        assert om.iteration_counts(theta[i]).max() <= 8
