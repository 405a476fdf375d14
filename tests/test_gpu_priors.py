"""GPU parity: the device prior transform (rvll_prior_batch) against the golden vectors produced by the
reference's own `.ppf` for every distribution of evidence/priors.py, plus the fused prior+log-L call."""
import zlib

import numpy as np
import pytest

import golden
import prior_cases as pc
from evidence_amd import GpuRVModel, priors as P
from evidence_amd.data import EpochTable
from evidence_amd.synthetic import make_workload

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180)]
Q, SETS = golden.prior_sets()


def one_param_model(spec):
    """A model with a single free parameter (an instrument offset) carrying the prior under test."""
    table = EpochTable.from_arrays(["ia"], [50000.0, 50001.0], [1.0, -1.0], [1.0, 1.0], [0, 0])
    return GpuRVModel({}, table, ["ia_offset"], priordict={"ia_offset": spec})


@pytest.mark.parametrize("name,args,vals,raised", SETS, ids=[f"{n}{tuple(a)}" for n, a, _, _ in SETS])
def test_device_ppf_matches_reference(gpu_required, name, args, vals, raised):
    with one_param_model(pc.spec_for(name, args)) as m:
        got = m.prior_transform_batch(Q.reshape(-1, 1))[:, 0]
    mask = pc.comparable_mask(name, Q, raised)
    err = pc.rel_err(got[mask], vals[mask], pc.abs_scale(name, args))
    assert err.max() <= pc.base_tol(name, args), (name, args, float(err.max()), float(Q[mask][err.argmax()]))
    assert np.all(np.isnan(got[raised]))            # where the reference raises ValueError we return NaN


def test_reference_unit_test_uniform(gpu_required):
    # tests/test_priors.py:13-15
    with one_param_model(P.Uniform(4, 6)) as m:
        assert m.prior_transform(np.array([0.5]))[0] == 5 and m.prior_transform(np.array([0.0]))[0] == 4
        assert m.prior_transform(np.array([1.0]))[0] == 6


def test_known_51peg_prior_vector(gpu_required):
    """prior(cube = 0.37) on the shipped 51Peg config."""
    z = np.load(golden.GOLDEN / "loglike_51peg.npz")
    table = EpochTable.from_arrays(["hamilton"], z["time"], z["vrad"], z["svrad"], z["inst_id"])
    pri = {"hamilton_jitter": P.Uniform(0.0, 50.0), "hamilton_offset": P.Uniform(-10, 10),
           "planet1_ecc": P.Beta(0.867, 3.03), "planet1_k1": P.Jeffreys(0.1, 100.0),
           "planet1_ma0": P.Uniform(0.0, 2 * np.pi), "planet1_omega": P.Uniform(0.0, 2 * np.pi),
           "planet1_period": P.UniformFrequency(1, 100)}
    want = [18.5, -2.5999999999999996, 0.11465599401597197, 1.288249551693134, 2.324778563656447,
            2.324778563656447, 1.5780337699226765]
    with GpuRVModel({"planet1_epoch": 51050.0}, table, list(pri), priordict=pri) as m:
        got = m.prior_transform(np.full(7, 0.37))
    assert pc.rel_err(got, want).max() <= 1e-13


def test_fused_prior_loglike_equals_two_calls(gpu_required):
    w = make_workload(3)
    cube = w.sample_cube(4096, seed=11)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        theta, logl = m.prior_loglike_batch(cube)
        theta2 = m.prior_transform_batch(cube)
        logl2 = m.log_likelihood_batch(theta2)
    assert np.array_equal(theta, theta2) and np.array_equal(logl, logl2)
    assert np.all((theta[:, w.parnames.index("planet1_ecc")] >= 0) & (theta[:, w.parnames.index("planet1_ecc")] <= 1))


@pytest.mark.parametrize("n", [1, 100, 777, 20000])
def test_one_launch_form_is_bit_identical_for_every_prior_family(gpu_required, n):
    """cube -> theta -> log-L in ONE launch (the prior transform runs in the log-L kernel's staging step)
    against prior kernel + log-L kernel, with every family of evidence/priors.py on some parameter: light,
    table, iterative (Beta, Gamma) and the sorted groups; small (zero-copy), ragged and large batches;
    host-buffer and device-resident entry points."""
    from evidence_amd import priors as P
    w = make_workload(3)
    fams = {
        "harps_jitter": P.TruncatedRayleigh(3.0, 30.0), "harps_offset": P.Normal(0.5, 4.0),
        "hires_jitter": P.Gamma(2.0, 0.7), "hires_offset": P.Binormal(-3.0, 1.0, 4.0, 2.0, 0.3),
        "planet1_ecc": P.Beta(0.867, 3.03), "planet1_k1": P.ModJeffreys(1.0, 80.0), "planet1_ma0": P.Sine(0.0, 180.0),
        "planet1_omega": P.Uniform(0.0, 6.283185307179586), "planet1_period": P.SortedLogUniform(1.5, 900.0),
        "planet2_ecc": P.TruncatedUNormal(0.1, 0.2, 0.0, 0.95), "planet2_k1": P.LogNormal(0.8, 0.0, 5.0),
        "planet2_ma0": P.AsymmetricNormal(3.0, 0.5, 1.5), "planet2_omega": P.PowerLaw(-0.5, 0.1, 6.0),
        "planet2_period": P.SortedLogUniform(1.5, 900.0),
        "planet3_ecc": P.Beta(2.0, 5.0), "planet3_k1": P.Alpha(3.0), "planet3_ma0": P.DoublePowerLaw(0.5, -1.5, 2.0, 0.5, 6.0),
        "planet3_omega": P.UniformFrequency(0.5, 6.0), "planet3_period": P.SortedLogUniform(1.5, 900.0),
    }
    assert sorted(fams) == list(w.parnames), sorted(set(w.parnames) ^ set(fams))
    cube = np.random.default_rng(n).random((n, w.ndim))
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=fams) as m:
        theta2 = m.prior_transform_batch(cube)
        logl2 = m.log_likelihood_batch(theta2)
        theta, logl = m.prior_loglike_batch(cube)                        # host buffers, one launch
        m.dev_upload_cube(cube)
        m.dev_prior_loglike(n)                                           # device resident, one launch
        th3, ll3, _ = m.dev_download(n, theta=True)
    same = lambda a, b: np.array_equal(a, b, equal_nan=True)
    assert same(theta, theta2) and same(logl, logl2)
    assert same(th3, theta2) and same(ll3, logl2)
    assert np.isfinite(theta).mean() > 0.99
    p = theta[:, [w.parnames.index(f"planet{k}_period") for k in (1, 2, 3)]]
    assert (np.diff(p, axis=1) >= 0).all()                                # the sorted group came out ordered


def test_prior_before_set_priors_is_an_error(gpu_required):
    from evidence_amd import RvllError
    w = make_workload(1)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
        with pytest.raises(RvllError):
            m.prior_transform_batch(w.sample_cube(4, 0))


def test_sorted_uniform_forced_identifiability(gpu_required):
    """pypolychord's SortedUniformPrior is absent from the reference checkout (parity unpinned);
    checked against the documented transform t[N-1]=x[N-1]^(1/N), t[n]=x[n]^(1/(n+1)) t[n+1]."""
    table = EpochTable.from_arrays(["ia"], [50000.0, 50001.0], [1.0, -1.0], [1.0, 1.0], [0, 0])
    names = ["planet1_period", "planet2_period", "planet3_period", "ia_offset"]
    fixed = {}
    for n in (1, 2, 3):
        fixed.update({f"planet{n}_k1": 1.0, f"planet{n}_ecc": 0.0, f"planet{n}_omega": 0.0, f"planet{n}_ma0": 0.0,
                      f"planet{n}_epoch": 0.0})
    pri = {"planet1_period": P.SortedUniform(1.0, 100.0), "planet2_period": P.SortedUniform(1.0, 100.0),
           "planet3_period": P.SortedUniform(1.0, 100.0), "ia_offset": P.Uniform(-1, 1)}
    rng = np.random.default_rng(0)
    cube = rng.random((500, 4))
    with GpuRVModel(fixed, table, names, priordict=pri) as m:
        idx = [m.parnames.index(f"planet{n}_period") for n in (1, 2, 3)]
        got = m.prior_transform_batch(cube)
    x = cube[:, idx]
    t = np.empty_like(x)
    t[:, 2] = x[:, 2] ** (1 / 3)
    t[:, 1] = x[:, 1] ** (1 / 2) * t[:, 2]
    t[:, 0] = x[:, 0] ** (1 / 1) * t[:, 1]
    want = 1.0 + 99.0 * t
    assert pc.rel_err(got[:, idx], want).max() <= 1e-14
    assert np.all(np.diff(got[:, idx], axis=1) >= 0)          # sorted: that is the point of the prior


@pytest.mark.parametrize("name,args,vals,raised", SETS, ids=[f"{n}{tuple(a)}" for n, a, _, _ in SETS])
def test_device_ppf_matches_oracle_on_dense_random_q(gpu_required, name, args, vals, raised):
    """Beyond the 64-point golden grid: 30 000 random unit-cube coordinates per distribution against the
    numpy/scipy oracle (itself pinned by the golden vectors, tests/test_priors_oracle.py)."""
    from oracle import priors_oracle as po
    rng = np.random.default_rng(zlib.crc32(f"{name}{tuple(args)}".encode()) % 100000)    # (str hashes change per process)
    q = np.concatenate([rng.random(28_000), 10.0 ** rng.uniform(-9, -1, 1000), 1 - 10.0 ** rng.uniform(-9, -1, 1000)])
    with one_param_model(pc.spec_for(name, args)) as m:
        got = m.prior_transform_batch(q.reshape(-1, 1))[:, 0]
    ref = po.ppf(name, args, q)
    mask = pc.comparable_mask(name, q, np.isnan(ref))
    err = pc.rel_err(got[mask], ref[mask], pc.abs_scale(name, args))
    over = err - pc.tolerance(name, args, ref[mask])
    assert over.max() <= 0, (name, args, float(err[over.argmax()]), float(q[mask][over.argmax()]))
    assert np.all(np.isnan(got[np.isnan(ref)]))


def test_sorted_loguniform_against_oracle(gpu_required):
    from oracle import priors_oracle as po
    table = EpochTable.from_arrays(["ia"], [50000.0, 50001.0], [1.0, -1.0], [1.0, 1.0], [0, 0])
    names = ["planet1_period", "planet2_period", "ia_offset"]
    fixed = {}
    for n in (1, 2):
        fixed.update({f"planet{n}_k1": 1.0, f"planet{n}_ecc": 0.0, f"planet{n}_omega": 0.0, f"planet{n}_ma0": 0.0,
                      f"planet{n}_epoch": 0.0})
    pri = {"planet1_period": P.SortedLogUniform(1.5, 1000.0), "planet2_period": P.SortedLogUniform(1.5, 1000.0),
           "ia_offset": P.Uniform(-1, 1)}
    cube = np.random.default_rng(1).random((2000, 3))
    with GpuRVModel(fixed, table, names, priordict=pri) as m:
        idx = [m.parnames.index("planet1_period"), m.parnames.index("planet2_period")]
        got = m.prior_transform_batch(cube)
    want = po.sorted_uniform(cube[:, idx], 1.5, 1000.0, log=True)
    assert pc.rel_err(got[:, idx], want).max() <= 1e-13


@pytest.mark.parametrize("umax", [None, 3.0, 0.0])
def test_slim_prior_stage_hands_over_what_its_tables_do_not_cover(gpu_required, umax):
    """The one-launch cube -> log-L kernel evaluates Beta / Gamma quantiles by their verified tables only
    (|logit q| <= 30) and leaves every other element — q = 0, q = 1, the far tails, anything when the range is
    lowered — to the prior kernels with the full solvers: whatever the route, the result is what
    prior_transform_batch + log_likelihood_batch return, bit for bit."""
    from evidence_amd import priors as P
    w = make_workload(3)
    pri = w.priordict()
    pri["hires_jitter"] = P.Gamma(2.0, 0.7)
    rng = np.random.default_rng(5)
    cube = rng.random((600, w.ndim))
    ecc = [w.parnames.index(f"planet{k}_ecc") for k in (1, 2, 3)]
    gam = w.parnames.index("hires_jitter")
    cube[0, ecc[0]] = 0.0
    cube[1, ecc[1]] = 1.0 - 2.0 ** -53
    cube[2, ecc[2]] = 1e-15
    cube[3, gam] = 1e-300
    cube[4, gam] = 1.0 - 1e-15
    cube[5, ecc[0]] = 1e-13
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=pri) as m:
        if umax is not None:
            m.set_slim_table_range(umax)
        theta2 = m.prior_transform_batch(cube)
        logl2 = m.log_likelihood_batch(theta2)
        for n in (6, 40, 600):                                           # zero-copy small batch, pinned staging, plain
            theta, logl = m.prior_loglike_batch(cube[:n])
            assert np.array_equal(theta, theta2[:n], equal_nan=True) and np.array_equal(logl, logl2[:n], equal_nan=True), n
            m.dev_upload_cube(cube[:n])
            m.dev_prior_loglike(n)
            th3, ll3, fl3 = m.dev_download(n, theta=True, flags=True)
            assert np.array_equal(th3, theta2[:n], equal_nan=True) and np.array_equal(ll3, logl2[:n], equal_nan=True), n
            assert not (fl3 & ~7).any()                                   # the internal deferral bit never leaves
        # and a batch the tables cover entirely still takes the one launch and agrees
        inner = rng.random((300, w.ndim))
        t4, l4 = m.prior_loglike_batch(inner)
        assert np.array_equal(l4, m.log_likelihood_batch(m.prior_transform_batch(inner)))
    assert np.isfinite(theta2[6:]).all()
