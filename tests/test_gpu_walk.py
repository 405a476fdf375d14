"""The device-resident slice-sampling walk (rvll_slice_walk, GpuRVModel.slice_walk): invariants of one walk,
uniformity without a constraint, the analytic Gaussian evidence, and the 51 Peg evidence against the host-driven
walk of the same sampler."""
import numpy as np
import pytest

from evidence_amd import GpuRVModel
from evidence_amd.callbacks import make_ultranest_callbacks, wrapped_params
from evidence_amd.nested import run_nested_slice
from evidence_amd.synthetic import make_workload

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def _start(m, w, k, seed, quantile=0.5):
    rng = np.random.default_rng(seed)
    cube = rng.random((k, m.ndim))
    theta, logl = m.prior_loglike_batch(cube)
    lstar = float(np.quantile(logl, quantile))
    keep = logl > lstar
    cube, theta, logl = cube[keep], theta[keep], logl[keep]
    d0 = cube - cube.mean(axis=0)
    chol = np.linalg.cholesky(d0.T @ d0 / (len(cube) - 1) + 1e-14 * np.eye(m.ndim))
    return cube, theta, logl, lstar, chol


@pytest.mark.parametrize("cfg, k", [(3, 3000), (1, 700), (5, 300)])
def test_walk_invariants(gpu_required, cfg, k):
    w = make_workload(cfg)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        cube, theta, logl, lstar, chol = _start(m, w, k, seed=cfg)
        wr = wrapped_params(m.parnames)
        c2, t2, l2, n = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=12, seed=11)
        th_chk, ll_chk = m.prior_loglike_batch(c2)
        c3, t3, l3, n3 = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=12, seed=11)
        c4, _, _, _ = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=12, seed=12)
        c0, t0, l0, n0 = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=0, seed=11)
    assert (l2 > lstar).all() and ((c2 >= 0) & (c2 < 1)).all()
    assert np.array_equal(th_chk, t2) and np.array_equal(ll_chk, l2)          # outputs describe the same points
    assert n >= 12 * len(cube) and np.mean(np.any(c2 != cube, axis=1)) > 0.99
    assert np.array_equal(c2, c3) and np.array_equal(l2, l3) and n == n3      # deterministic for a seed
    assert not np.array_equal(c2, c4)
    assert np.array_equal(c0, cube) and np.array_equal(l0, logl) and n0 == 0   # nsteps = 0: nothing moves


@pytest.mark.parametrize("precision", ["mixed", "fp32"])
def test_walk_and_scalar_server_in_the_reduced_precision_modes(gpu_required, precision):
    """The walk kernel and the scalar-call server are instantiated per precision mode: same invariants, judged
    against the batch path of the SAME mode — theta bit for bit (the prior transform is fp64 in every mode), log-L to float
    rounding of the model: the reduced modes' Newton loop runs wave-wide since round 4 and an item's step count goes by its
    wave (tests/test_gpu_forms.py)."""
    w = make_workload(3)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict(), precision=precision) as m:
        cube, theta, logl, lstar, chol = _start(m, w, 2000, seed=4)
        c2, t2, l2, n = m.slice_walk(cube, theta, logl, lstar, chol, wrapped_params(m.parnames), nsteps=6, seed=2)
        th_chk, ll_chk = m.prior_loglike_batch(c2)
        assert (l2 > lstar).all() and np.array_equal(th_chk, t2) and n >= 6 * len(cube)
        assert np.max(np.abs(ll_chk - l2) / np.abs(l2)) <= 3e-7
        want = m.log_likelihood_batch(t2[:32])
        m.scalar_server(True)
        got = np.array([m.log_likelihood(x) for x in t2[:32]])
        assert np.max(np.abs(got - want) / np.abs(want)) <= 3e-7


def test_walk_gives_up_a_move_after_max_rounds_and_accepts_no_wrapping(gpu_required):
    """A tight constraint with max_rounds = 1: most candidates are rejected and the move is given up — the walker
    stays where it was, still above lstar; ncalls counts exactly the one candidate per walker.  wrapped=None works."""
    w = make_workload(3)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        cube, theta, logl, lstar, chol = _start(m, w, 4000, seed=9, quantile=0.97)
        c2, t2, l2, n = m.slice_walk(cube, theta, logl, lstar, chol, None, nsteps=1, max_rounds=1, seed=3)
        th_chk, ll_chk = m.prior_loglike_batch(c2)
    assert n == len(cube)
    assert (l2 > lstar).all() and np.array_equal(ll_chk, l2) and np.array_equal(th_chk, t2)
    stayed = np.all(c2 == cube, axis=1)
    assert 0.05 < stayed.mean() < 0.95                       # the one candidate was rejected for these
    assert np.array_equal(l2[stayed], logl[stayed]) and np.array_equal(t2[stayed], theta[stayed])


def test_unconstrained_walk_is_uniform_in_the_cube(gpu_required):
    """lstar = -inf: every first candidate is accepted, so a long walk must forget its start and fill the unit
    cube uniformly — walls, circular parameters and the chord logic included."""
    w = make_workload(2)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        k = 20000
        cube = np.full((k, m.ndim), 0.31)
        theta, logl = m.prior_loglike_batch(cube)
        wr = wrapped_params(m.parnames)
        c, t, l, n = m.slice_walk(cube, theta, logl, -np.inf, np.eye(m.ndim), wr, nsteps=40, seed=5)
    assert n == 40 * k
    assert np.abs(c.mean(axis=0) - 0.5).max() < 0.01
    assert np.abs(c.var(axis=0) - 1.0 / 12.0).max() < 0.004
    assert np.abs(np.corrcoef(c.T) - np.eye(m.ndim)).max() < 0.03
    hist = np.stack([np.histogram(c[:, j], bins=10, range=(0, 1))[0] for j in range(m.ndim)])
    assert np.abs(hist / k - 0.1).max() < 0.012


def test_gaussian_evidence_with_the_walk_on_the_device(gpu_required):
    """Two free offsets, one unit-variance datum each, no planet: log-L = -(a^2 + b^2)/2 - ln 2pi, so with
    Uniform(-10, 10) priors ln Z = -ln 400 = -5.9915 (the reference's 2-D Gaussian known answer, shifted by
    the normalisation: tests/test_polychord.py:139)."""
    from evidence_amd import priors as P
    from evidence_amd.data import EpochTable
    table = EpochTable.from_arrays(["a", "b"], [1.0, 2.0], [0.0, 0.0], [1.0, 1.0], [0, 1])
    pri = {"a_offset": P.Uniform(-10, 10), "b_offset": P.Uniform(-10, 10)}
    with GpuRVModel({}, table, list(pri), priordict=pri) as m:
        prior, loglike = make_ultranest_callbacks(m, vectorized=True)
        out = [run_nested_slice(prior, loglike, 2, nlive=1000, dlogz=0.01, seed=s, walker=m.slice_walk,
                                nsteps=10, max_calls=20_000_000) for s in (1, 2, 3)]
        # ... and with the live set itself resident on the device (whitening summed there too)
        out += [run_nested_slice(None, None, 2, nlive=1000, dlogz=0.01, seed=s, live=m, nsteps=10, max_calls=20_000_000)
                for s in (4, 5, 6)]
    for r in out:
        assert abs(r.logz - (-np.log(400.0))) < 4 * r.logzerr + 0.05, (r.logz, r.logzerr)
    assert abs(np.mean([r.logz for r in out]) + np.log(400.0)) < 0.12


def test_device_walk_is_unbiased_on_a_4d_gaussian(gpu_required):
    """Many seeds on a likelihood with an analytic evidence (four free offsets, one unit-variance datum each:
    ln Z = -4 ln 20): the mean over seeds must sit on the truth within its standard error — a biased walk
    (wrong chord, wrong acceptance, correlated random numbers) shows up here, a single run would hide it."""
    from evidence_amd import priors as P
    from evidence_amd.data import EpochTable
    names = ["a", "b", "c", "d"]
    table = EpochTable.from_arrays(names, [1.0, 2.0, 3.0, 4.0], [0.3, -0.2, 0.1, 0.0], [1.0] * 4, [0, 1, 2, 3])
    pri = {f"{n}_offset": P.Uniform(-10, 10) for n in names}
    truth = -4 * np.log(20.0)
    with GpuRVModel({}, table, list(pri), priordict=pri) as m:
        prior, loglike = make_ultranest_callbacks(m, vectorized=True)
        z = np.array([run_nested_slice(prior, loglike, 4, nlive=1000, kbatch=500, dlogz=0.01, nsteps=12, seed=s,
                                       walker=m.slice_walk, max_calls=50_000_000).logz for s in range(1, 25)])
    se = z.std(ddof=1) / np.sqrt(len(z))
    assert abs(z.mean() - truth) < 4 * se + 0.01, (z.mean() - truth, se)
    assert z.std() < 0.15                                   # ~ sqrt(H / nlive) with H ~ 6


def test_51peg_evidence_device_walk_agrees_with_host_walk(gpu_required):
    from pathlib import Path
    from evidence_amd.config import read_config
    cfg = Path(__file__).resolve().parents[1] / "examples" / "51peg" / "config_51peg.py"
    rundict, datadict, priordict, fixed = read_config(cfg, nplanets=1)
    with GpuRVModel(fixed, datadict, list(priordict), priordict=priordict) as m:
        prior, loglike = make_ultranest_callbacks(m, vectorized=True)
        wrap = wrapped_params(m.parnames)
        kw = dict(nlive=400, dlogz=0.5, wrapped=wrap, max_calls=8_000_000)
        host = run_nested_slice(prior, loglike, m.ndim, seed=1, prior_loglike=m.prior_loglike_batch, **kw)
        dev = [run_nested_slice(prior, loglike, m.ndim, seed=s, walker=m.slice_walk, **kw) for s in (1, 2)]
    for d in dev:
        # on this sharply multimodal posterior the sampler scatters by ~1.5 in ln Z from seed to seed at 400 live
        # points, with either walk (profiles/r01_walk_bias_check.txt has the unimodal, unbiased case)
        assert abs(d.logz - host.logz) < 5 * np.hypot(d.logzerr, host.logzerr) + 3.0, (d.logz, host.logz)
        wgt = np.exp(d.logwt)
        ip, ik = m.parnames.index("planet1_period"), m.parnames.index("planet1_k1")
        assert abs(np.sum(wgt * d.samples[:, ip]) - 4.2308) < 0.01
        assert abs(np.sum(wgt * d.samples[:, ik]) - 56.0) < 6.0


def test_slim_walk_with_fat_finish_is_the_fat_walk(gpu_required, monkeypatch):
    """The walk's fast instantiation evaluates Beta / Gamma quantiles by their verified tables only; a walker whose
    candidate needs more stops at the start of that move and the instantiation with the full solvers finishes it,
    retracing the interrupted move from the same counter-based random numbers.  Whatever share of the walkers takes
    that route — none (default range), about half of the candidates (|logit q| <= 1), all of them (range 0) — the
    end points are those of the full-solver walk alone."""
    w = make_workload(3)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        cube, theta, logl, lstar, chol = _start(m, w, 1500, seed=21)
        wr = wrapped_params(m.parnames)
        monkeypatch.setenv("RVLL_WALK_FAT", "1")
        ref = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=9, seed=4)
        monkeypatch.delenv("RVLL_WALK_FAT")
        got = {}
        for umax in (30.0, 1.0, 0.0):
            m.set_slim_table_range(umax)
            got[umax] = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=9, seed=4)
    for umax, g in got.items():
        assert np.array_equal(g[0], ref[0]) and np.array_equal(g[1], ref[1]) and np.array_equal(g[2], ref[2]), umax
    assert got[30.0][3] == ref[3]                 # nothing deferred: the same number of likelihood calls
    assert got[0.0][3] >= ref[3]                  # everything deferred at its first candidate, then redone


def test_sharded_walk_draws_what_the_unsharded_walk_draws(gpu_required):
    """walker_base: rows [lo, hi) walked on their own with walker_base = lo end where the full walk puts them —
    the counters of the random numbers are (seed, walker_base + row, move, draw) — so a multi-GPU run
    (sharded.ShardedWalker) is reproducible for a seed whatever the number of ranks."""
    from evidence_amd.sharded import partition
    w = make_workload(3)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        cube, theta, logl, lstar, chol = _start(m, w, 900, seed=31)
        wr = wrapped_params(m.parnames)
        full = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=7, seed=6)
        for world in (2, 3):
            parts = [m.slice_walk(cube[lo:hi], theta[lo:hi], logl[lo:hi], lstar, chol, wr, nsteps=7, seed=6, walker_base=lo)
                     for lo, hi in partition(len(cube), world)]
            assert np.array_equal(np.concatenate([p[0] for p in parts]), full[0]), world
            assert np.array_equal(np.concatenate([p[2] for p in parts]), full[2]), world
            assert sum(p[3] for p in parts) == full[3]
        other = m.slice_walk(cube[:100], theta[:100], logl[:100], lstar, chol, wr, nsteps=7, seed=6, walker_base=100)
    assert not np.array_equal(other[0], full[0][:100])            # other rows, other draws


def test_speculation_and_the_walker_queue_change_nothing_in_the_results(gpu_required, monkeypatch):
    """Two things about HOW the walk is scheduled must not show in WHAT it returns: free tile slots evaluating a
    walker's next candidates ahead (rvll_set_walk_speculation; consumed in the order the walker would have met them),
    and freed walker slots drawing the remaining rows from a queue (the launch holds no more workgroups than the chip
    does).  End points, log-L and the number of likelihood calls are those of one candidate per walker and iteration
    with one workgroup per PB rows — also when walkers are deferred to the full-solver instantiation on the way."""
    w = make_workload(3)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        cube, theta, logl, lstar, chol = _start(m, w, 2600, seed=41, quantile=0.8)
        wr = wrapped_params(m.parnames)
        m.set_points_per_block(8)                                  # (a walk this small would get one walker per workgroup)
        runs = {}
        for umax in (30.0, 1.0):                                   # nobody deferred / about half of the candidates
            m.set_slim_table_range(umax)
            for queue in ("0", "1", "3", None):                    # off / 1 CU's worth of workgroups / 3 / the whole chip
                if queue is None:
                    monkeypatch.delenv("RVLL_WALK_QUEUE", raising=False)
                else:
                    monkeypatch.setenv("RVLL_WALK_QUEUE", queue)
                for ahead in (1, 2, 4, 8):
                    m.set_walk_speculation(ahead)
                    out = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=9, seed=5)
                    runs[(umax, queue, ahead)] = out + (m.slice_walk_evaluated(),)
        # the full-solver instantiation through the queue and the two-part launch as well
        m.set_slim_table_range(30.0)
        m.set_walk_speculation(4)
        monkeypatch.setenv("RVLL_WALK_FAT", "1")
        monkeypatch.setenv("RVLL_WALK_QUEUE", "1")
        fat_queue = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=9, seed=5)
        monkeypatch.setenv("RVLL_WALK_PARTS", "1")
        fat_one_part = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=9, seed=5)
        monkeypatch.delenv("RVLL_WALK_FAT"); monkeypatch.delenv("RVLL_WALK_QUEUE"); monkeypatch.delenv("RVLL_WALK_PARTS")
    for got in (fat_queue, fat_one_part):
        ref = runs[(30.0, "0", 1)]
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]) and got[3] == ref[3]
    for umax in (30.0, 1.0):
        ref = runs[(umax, "0", 1)]
        # no speculation: every evaluated slot is a call — except the candidates of moves a deferral interrupted, which
        # the slim pass evaluated and the full-solver pass evaluates (and counts) again
        assert ref[4] == ref[3] if umax == 30.0 else ref[4] > ref[3]
        for key, got in runs.items():
            if key[0] != umax:
                continue
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]), key
            assert got[3] == ref[3], key
            assert got[4] >= got[3] and (key[2] > 1 or umax != 30.0 or got[4] == got[3]), key
    assert runs[(30.0, "1", 4)][4] > runs[(30.0, "1", 4)][3]       # speculation did evaluate candidates ahead
    assert np.array_equal(runs[(30.0, "0", 1)][0], runs[(1.0, "0", 1)][0])      # and deferral changes nothing either
    # ... not even the count: a move interrupted by a deferral is retraced by the full-solver pass from its first candidate,
    # and its candidates are counted there only (they used to be counted twice)
    assert runs[(1.0, "0", 1)][3] == runs[(30.0, "0", 1)][3] == fat_queue[3]


def test_rows_form_of_the_second_part_is_the_queue_walk(gpu_required, monkeypatch):
    """With more rows than walker slots the rest of the walk after its first moves can run in the "rows" form (RVLL_WALK_ROWS=1;
    measured 7 % slower than the queue at cfg3 and therefore not the default, profiles/r03_walk_forms.txt): every workgroup
    owns an equal share of the rows (by what they cost so far), parks them in LDS and interleaves them over its walker
    slots at move boundaries (rvll_walk.hip, slice_walk_rows_kernel).  Which slot walks which move of which row changes
    nothing: end points, theta, log-L and the call count are those of the queue form and of one launch — with rows
    deferred to the full-solver pass on the way, with a share larger than one launch's LDS holds (several launches), and
    with fewer rows than a workgroup has slots."""
    w = make_workload(3)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        wr = wrapped_params(m.parnames)
        m.set_points_per_block(8)
        for k, cus in ((3000, "1"), (1000, "2"), (70, "1")):     # 4 workgroups x 750 rows (chunked) / 8 x 125 / 4 x 18
            cube, theta, logl, lstar, chol = _start(m, w, k, seed=60 + k, quantile=0.8)
            monkeypatch.setenv("RVLL_WALK_QUEUE", cus)
            for umax in (30.0, 1.0):
                m.set_slim_table_range(umax)
                runs = {}
                for name, env in (("rows", {"RVLL_WALK_ROWS": "1"}), ("cu", {"RVLL_WALK_ROWS": "2"}), ("half", {"RVLL_WALK_ROWS": "3"}), ("queue", {}), ("one", {"RVLL_WALK_PARTS": "1"})):
                    for key, val in env.items():
                        monkeypatch.setenv(key, val)
                    runs[name] = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=17, seed=8) + (m.slice_walk_evaluated(),)
                    for key in env:
                        monkeypatch.delenv(key)
                for name in ("rows", "cu", "half", "queue"):
                    got, ref = runs[name], runs["one"]
                    assert all(np.array_equal(a, b) for a, b in zip(got[:3], ref[:3])) and got[3] == ref[3], (k, umax, name)
                    assert got[4] >= got[3]
            m.set_slim_table_range(30.0)
            th_chk, ll_chk = m.prior_loglike_batch(runs["rows"][0])
            assert np.array_equal(th_chk, runs["rows"][1]) and np.array_equal(ll_chk, runs["rows"][2])
            assert (runs["rows"][2] > lstar).all()


def test_rounds_form_is_the_single_kernel_walk(gpu_required, monkeypatch):
    """The walk has a second form (round 4): a sequence of ROUNDS — per group of walkers one launch that accepts / shrinks,
    proposes and prior-transforms (directions of all moves made ahead of time), one launch of the batch theta -> log-L tiles
    over the group's compacted candidates; walker state resident in HBM (csrc/rvll_rounds.h) — instead of one kernel that
    keeps its walkers in LDS.  It is what walks of 6144 .. 24576 rows take by default (where it was measured faster,
    profiles/r04_rounds_sizes.txt); RVLL_WALK_ROUNDS=1 / 0 forces / forbids it.  How the walkers are grouped, how the rounds
    are issued (a stream per group, chained or not), which
    form the tiles take, how many candidates a walker gets ahead, how deep the host keeps the queues: none of it shows in
    the results, which are those of the single-kernel walk bit for bit — end points, theta, log-L, the number of likelihood
    calls — with and without walkers deferred to the full-solver pass on the way."""
    knobs = ("RVLL_WALK_ROUNDS", "RVLL_ROUNDS_GROUPS", "RVLL_ROUNDS_FREE", "RVLL_ROUNDS_DEPTH", "RVLL_WALK_SPEC", "RVLL_ROUNDS_FORM",
             "RVLL_ROUNDS_MODE", "RVLL_ROUNDS_CHAIN", "RVLL_ROUNDS_PRIO", "RVLL_ROUNDS_W", "RVLL_ROUNDS_PB")
    variants = [{}, {"RVLL_ROUNDS_GROUPS": "1"}, {"RVLL_ROUNDS_GROUPS": "2", "RVLL_ROUNDS_FORM": "cu"}, {"RVLL_ROUNDS_GROUPS": "4", "RVLL_WALK_SPEC": "1"},
                {"RVLL_ROUNDS_GROUPS": "2"}, {"RVLL_ROUNDS_CHAIN": "1"}, {"RVLL_ROUNDS_PRIO": "1", "RVLL_ROUNDS_W": "16"},
                {"RVLL_WALK_SPEC": "16", "RVLL_ROUNDS_FREE": "100000"}, {"RVLL_ROUNDS_FREE": "1", "RVLL_ROUNDS_DEPTH": "1"},
                {"RVLL_ROUNDS_DEPTH": "9", "RVLL_ROUNDS_PB": "3"}]
    for cfg, k, nsteps in ((3, 5000, 9), (3, 131, 21), (1, 40, 7), (5, 600, 5)):
        w = make_workload(cfg)
        with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
            cube, theta, logl, lstar, chol = _start(m, w, k, seed=70 + cfg, quantile=0.6)
            wr = wrapped_params(m.parnames)
            for umax in ((30.0, 1.0) if cfg == 3 else (30.0,)):
                m.set_slim_table_range(umax)
                monkeypatch.setenv("RVLL_WALK_ROUNDS", "0")
                ref = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=nsteps, seed=13)
                assert m.slice_walk_rounds() == 0
                for env in variants:
                    monkeypatch.setenv("RVLL_WALK_ROUNDS", "1")
                    for key, val in env.items():
                        monkeypatch.setenv(key, val)
                    got = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=nsteps, seed=13)
                    rounds, slots = m.slice_walk_rounds(), m.slice_walk_evaluated()
                    for key in env:
                        monkeypatch.delenv(key)
                    assert all(np.array_equal(a, b) for a, b in zip(got[:3], ref[:3])) and got[3] == ref[3], (cfg, k, umax, env)
                    assert rounds >= (nsteps if umax == 30.0 else 1) and slots >= got[3], (cfg, k, umax, env)       # it did walk in rounds
                    if env.get("RVLL_WALK_SPEC") == "1" and umax == 30.0:
                        assert slots == got[3]                                             # no candidates ahead: every slot a call
                monkeypatch.delenv("RVLL_WALK_ROUNDS")
            m.set_slim_table_range(30.0)
            th_chk, ll_chk = m.prior_loglike_batch(ref[0])
            assert np.array_equal(th_chk, ref[1]) and np.array_equal(ll_chk, ref[2]) and (ref[2] > lstar).all()
    # which form a walk takes by default goes by its size
    w = make_workload(3)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        cube, theta, logl, lstar, chol = _start(m, w, 16000, seed=77, quantile=0.5)
        wr = wrapped_params(m.parnames)
        big = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=5, seed=3)
        assert m.slice_walk_rounds() >= 5 and len(cube) >= 6144
        small = m.slice_walk(cube[:3000], theta[:3000], logl[:3000], lstar, chol, wr, nsteps=5, seed=3)
        assert m.slice_walk_rounds() == 0
        monkeypatch.setenv("RVLL_WALK_ROUNDS", "0")
        ref = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=5, seed=3)
        monkeypatch.delenv("RVLL_WALK_ROUNDS")
        assert all(np.array_equal(a, b) for a, b in zip(big[:3], ref[:3])) and big[3] == ref[3]
        assert all(np.array_equal(a[:3000], b) for a, b in zip(ref[:3], small[:3]))       # (rows walk alike whatever walks beside them)
    for key in knobs:
        assert key not in __import__("os").environ


def test_rounds_form_with_few_candidates_a_move(gpu_required, monkeypatch):
    """max_rounds = 1 and 3 (a move is given up after that many candidates) through the rounds form at its default queue depth and at
    deeper ones: the host keeps several rounds queued beyond the last one it has seen start, and its bound on the number of
    rounds has to allow for them — it refused such walks while their last rounds were still in the queue (found by
    scripts/walk_soak.py in round 4).  Same results as the single-kernel walk, bit for bit."""
    w = make_workload(3)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        cube, theta, logl, lstar, chol = _start(m, w, 3000, seed=91, quantile=0.7)
        wr = wrapped_params(m.parnames)
        for max_rounds, nsteps in ((1, 20), (3, 11), (1, 1)):
            monkeypatch.setenv("RVLL_WALK_ROUNDS", "0")
            ref = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=nsteps, max_rounds=max_rounds, seed=5)
            for depth in (None, "1", "9"):
                monkeypatch.setenv("RVLL_WALK_ROUNDS", "1")
                if depth:
                    monkeypatch.setenv("RVLL_ROUNDS_DEPTH", depth)
                got = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=nsteps, max_rounds=max_rounds, seed=5)
                assert m.slice_walk_rounds() > 0
                if depth:
                    monkeypatch.delenv("RVLL_ROUNDS_DEPTH")
                assert all(np.array_equal(a, b) for a, b in zip(got[:3], ref[:3])) and got[3] == ref[3], (max_rounds, nsteps, depth)
            monkeypatch.delenv("RVLL_WALK_ROUNDS")


def test_queue_serves_rows_with_nothing_left_to_do_and_a_ragged_last_workgroup(gpu_required, monkeypatch):
    """Sizes around the queue's edges: fewer walkers than one workgroup holds, a walker count that is not a multiple of
    the group size, one workgroup serving every row."""
    w = make_workload(1)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        wr = wrapped_params(m.parnames)
        for k in (7, 37, 530):
            cube, theta, logl, lstar, chol = _start(m, w, k, seed=50 + k)
            monkeypatch.setenv("RVLL_WALK_QUEUE", "0")
            ref = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=5, seed=9)
            for pb in (1, 5, 0):
                m.set_points_per_block(pb)
                monkeypatch.setenv("RVLL_WALK_QUEUE", "1")
                got = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=5, seed=9)
                assert all(np.array_equal(a, b) for a, b in zip(got[:3], ref[:3])) and got[3] == ref[3], (k, pb)
            th_chk, ll_chk = m.prior_loglike_batch(ref[0])
            assert np.array_equal(th_chk, ref[1]) and np.array_equal(ll_chk, ref[2])


def test_resident_live_set_is_the_host_managed_run_bit_for_bit(gpu_required):
    """The live set kept on the device (rvll_live_*; nested.run_nested_slice(live=model)): with the whitening factor
    computed on the host from a mirror of the rows (live_chol="host") the run is the host-managed one (walker=
    model.slice_walk) bit for bit — evidence, call count, every sample, weight and log-L, dead points in the order they
    died — because the same kernels walk the same rows with the same counter-based random numbers; only where the rows
    live between iterations differs."""
    w = make_workload(3)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        prior, loglike = make_ultranest_callbacks(m, vectorized=True)
        wr = wrapped_params(m.parnames)
        kw = dict(nlive=3000, kbatch=1000, nsteps=9, dlogz=1e-9, max_calls=400_000, wrapped=wr, seed=4)
        ref = run_nested_slice(prior, loglike, m.ndim, walker=m.slice_walk, **kw)
        got = run_nested_slice(None, None, m.ndim, live=m, live_chol="host", **kw)
        dev = run_nested_slice(None, None, m.ndim, live=m, **kw)                 # whitening summed on the device
        # the rows the device holds at the end are the run's live points; the dead store holds the rest
        u, theta, logl = m.live_get()
        th2, ll2 = m.prior_loglike_batch(u)
    assert ref.niter >= 3000 and got.niter == ref.niter and got.ncall == ref.ncall
    assert got.logz == ref.logz and got.information == ref.information
    assert np.array_equal(got.samples, ref.samples) and np.array_equal(got.logl, ref.logl) and np.array_equal(got.logwt, ref.logwt)
    assert np.array_equal(th2, theta) and np.array_equal(ll2, logl)
    assert np.array_equal(dev.samples[-3000:], theta) and dev.samples.shape == ref.samples.shape
    # the device's covariance differs from numpy's in the last bits (another summation order), so that run takes another
    # path through the same posterior: same size, same answer within the sampler's own scatter
    assert dev.niter == ref.niter and abs(dev.ncall - ref.ncall) < 0.1 * ref.ncall
    assert abs(dev.logz - ref.logz) < 5 * np.hypot(dev.logzerr, ref.logzerr) + 1.0


def test_live_step_api_contract(gpu_required):
    """rvll_live_step by itself: dying rows go to the dead store in order, walkers start from the given rows, end above
    lstar, replace the dying rows; the device's whitening factor is the Cholesky factor of the surviving rows' covariance."""
    w = make_workload(2)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        rng = np.random.default_rng(3)
        n, k = 900, 300
        cube = rng.random((n, m.ndim))
        logl = m.live_init(cube)
        theta0, ll0 = m.prior_loglike_batch(cube)
        assert np.array_equal(logl, ll0)
        order = np.argsort(logl, kind="stable")
        lstar = logl[order[k - 1]]
        start = order[k:][rng.integers(0, n - k, k)]
        wr = wrapped_params(m.parnames)
        new, used, chol = m.live_step(order, k, start, lstar, wr, nsteps=6, seed=12, return_chol=True)
        u, theta, ll = m.live_get()
        dth, dll = m.live_dead()
        # the same walk through the host-buffer entry point, from the same start rows with the same factor
        c2, t2, l2, used2 = m.slice_walk(cube[start], theta0[start], logl[start], lstar, chol, wr, nsteps=6, seed=12)
    assert used == used2 and np.array_equal(new, l2) and (new > lstar).all()
    assert np.array_equal(u[order[:k]], c2) and np.array_equal(theta[order[:k]], t2) and np.array_equal(ll[order[:k]], l2)
    keep = order[k:]
    assert np.array_equal(u[keep], cube[keep]) and np.array_equal(ll[keep], logl[keep])       # survivors untouched
    assert np.array_equal(dth, theta0[order[:k]]) and np.array_equal(dll, logl[order[:k]])
    d0 = cube[keep] - cube[keep].mean(axis=0)
    want = np.linalg.cholesky(d0.T @ d0 / (len(keep) - 1) + 1e-14 * np.eye(m.ndim))
    assert np.allclose(chol, want, rtol=1e-10, atol=1e-13) and np.allclose(np.triu(chol, 1), 0.0)
    # the order on the device (rvll_live_sort, round 4): the same step from ranks among the survivors instead of an order and rows
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        logl2 = m.live_init(cube)
        dl, lstar2, top = m.live_sort(k)
        assert np.array_equal(dl, logl[order[:k]]) and lstar2 == lstar and top == logl.max() and np.array_equal(logl2, logl)
        ranks = np.searchsorted(order[k:], start) if False else np.array([int(np.flatnonzero(order[k:] == s)[0]) for s in start])
        new2, used3 = m.live_step(None, k, ranks, lstar2, wr, nsteps=6, seed=12, chol=chol)
        u2, theta2, ll2 = m.live_get()
        assert used3 == used and np.array_equal(new2, new) and np.array_equal(u2, u) and np.array_equal(theta2, theta) and np.array_equal(ll2, ll)
        with pytest.raises(Exception, match="rvll_live_sort"):       # the order is used up: a second step needs a second sort
            m.live_step(None, k, ranks, lstar2, wr, nsteps=6, seed=12, chol=chol)
        ties = np.round(cube[:, :1] * 4) / 4 + 0 * cube            # many equal log-L values: ties go by row, as numpy's stable argsort
        lt = m.live_init(np.clip(ties, 0.0, 0.999))
        dl_t, _, _ = m.live_sort(k)
        assert np.array_equal(dl_t, lt[np.argsort(lt, kind="stable")[:k]])
    # a step that fails leaves the run as it was: a dying row listed twice is refused before anything is touched; a
    # covariance that cannot be factored (a NaN among the surviving rows) fails AFTER the dying rows were copied to the dead
    # store — which must not count them, or a retry would append them again
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        rng = np.random.default_rng(5)
        cube = rng.random((600, m.ndim))
        logl = m.live_init(cube)
        order = np.argsort(logl, kind="stable")
        start = order[200:][rng.integers(0, 400, 200)]
        twice = order.copy(); twice[1] = twice[0]
        with pytest.raises(Exception, match="listed twice"):
            m.live_step(twice, 200, start, logl[order[199]], wrapped_params(m.parnames), nsteps=3, seed=1)
        assert m.live_dead_count() == 0
        new, used = m.live_step(order, 200, start, logl[order[199]], wrapped_params(m.parnames), nsteps=3, seed=1)
        assert m.live_dead_count() == 200
        bad = cube.copy(); bad[7, 2] = np.nan
        logl_bad = m.live_init(bad)                                   # (a new run: the dead store is empty again)
        assert m.live_dead_count() == 0
        fin = np.where(np.isfinite(logl_bad), logl_bad, -1e300)
        order = np.argsort(fin, kind="stable")                        # the NaN row dies first only if listed first: keep it among the survivors
        order = np.concatenate([order[order != 7], [7]])
        with pytest.raises(Exception, match="positive definite"):
            m.live_step(order, 100, order[100:200], fin[order[99]], wrapped_params(m.parnames), nsteps=3, seed=1)
        assert m.live_dead_count() == 0                               # nothing was appended by the step that failed
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        with pytest.raises(RuntimeError, match="live_init"):
            m.live_step(np.arange(5), 2, np.array([3, 4]), 0.0)
        with pytest.raises(Exception, match="rvll_live_init"):       # ... and the C-ABI says the same when asked directly
            m._live_n = 5
            m.live_step(np.arange(5), 2, np.array([3, 4]), 0.0)


def test_converged_evidence_resident_against_host_managed(gpu_required):
    """Evidence runs to dlogz = 0.5 (the reference's UltraNest default, evidence/ultranest/__init__.py:333-338; the style of
    check of /root/reference/tests/test_polychord.py:75-151) at BASELINE configs[1] (one eccentric planet, 200 epochs): the
    run with the live set resident on the device and the host-managed run from the same seed agree within three combined
    sigma, and both stop because the live points' remaining mass is below dlogz, not at a call budget.  (cfg3's three
    exchangeable planets make ITS evidence a matter of which of the 3! modes a run keeps: profiles/r04_converged_runs.txt.)"""
    w = make_workload(2)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        prior, loglike = make_ultranest_callbacks(m, vectorized=True)
        wr = wrapped_params(m.parnames)
        for seed in (1, 2):
            kw = dict(nlive=2048, kbatch=512, dlogz=0.5, max_calls=200_000_000, wrapped=wr, seed=seed)
            res = run_nested_slice(None, None, m.ndim, live=m, **kw)
            host = run_nested_slice(prior, loglike, m.ndim, walker=m.slice_walk, prior_loglike=m.prior_loglike_batch, **kw)
            assert res.ncall < 100_000_000 and host.ncall < 100_000_000                     # converged, not truncated
            assert 0.05 < res.logzerr < 0.3 and 20 < res.information < 40
            assert abs(res.logz - host.logz) < 3 * np.hypot(res.logzerr, host.logzerr), (seed, res.logz, host.logz, res.logzerr)
            assert -540 < res.logz < -532


def test_walkers_that_end_on_a_wandering_solve_carry_the_exact_log_l(gpu_required, monkeypatch):
    """Round 4: a Kepler solve that wanders (e >= 0.97) is redone with correctly rounded sin / cos by every batch path
    (rvll_set_wander_exact) — but not inside the walk's tiles, where one such solve would hold a whole round up: the walk
    judges its candidates by the ordinary evaluation and puts the log-L of the points it ENDS on right afterwards.  On a
    model whose first planet lives at e = 0.95 .. 0.9925 (most candidates wander): every end point's log-L is the batch
    path's, bit for bit; the rounds form and the single-kernel form agree; and with the switch off the walk ends on the
    same positions with log-L values that differ on the wandering points by parts in 1e10."""
    import golden
    from evidence_amd import priors as P
    case = golden.high_ecc_case()
    lo, hi = case.theta.min(axis=0), case.theta.max(axis=0)
    pri = {n: P.Uniform(float(a), float(b if b > a else a + 1.0)) for n, a, b in zip(case.parnames, lo, hi)}
    pri["planet1_ecc"] = P.Uniform(0.95, 0.9925)
    out = {}
    for exact in (True, False):
        with GpuRVModel(case.fixed, case.table, case.parnames, priordict=pri) as m:
            m.set_wander_exact(exact)
            rng = np.random.default_rng(5)
            cube = rng.random((6000, m.ndim))
            theta, logl = m.prior_loglike_batch(cube)
            lstar = float(np.quantile(logl, 0.5))
            keep = logl > lstar
            cube, theta, logl = cube[keep], theta[keep], logl[keep]
            d0 = cube - cube.mean(axis=0)
            chol = np.linalg.cholesky(d0.T @ d0 / (len(cube) - 1) + 1e-14 * np.eye(m.ndim))
            wr = wrapped_params(m.parnames)
            for form in ("0", "1"):
                monkeypatch.setenv("RVLL_WALK_ROUNDS", form)
                c2, t2, l2, n = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=6, seed=3)
                th_chk, ll_chk = m.prior_loglike_batch(c2)
                assert np.array_equal(th_chk, t2) and np.array_equal(ll_chk, l2), (exact, form)
                out[(exact, form)] = (c2, l2, n)
            monkeypatch.delenv("RVLL_WALK_ROUNDS")
            out[(exact, "flags")] = m.log_likelihood_batch(m.prior_transform_batch(out[(exact, "0")][0]), return_flags=True)[1]
    for exact in (True, False):
        a, b = out[(exact, "0")], out[(exact, "1")]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    on, off = out[(True, "0")], out[(False, "0")]
    assert np.array_equal(on[0], off[0]) and on[2] == off[2]                 # the same walk ...
    wandered = (out[(True, "flags")] & 4) != 0
    assert wandered.mean() > 0.2 and np.array_equal(on[1][~wandered], off[1][~wandered])
    diff = np.abs(on[1] - off[1]) / np.abs(on[1])
    assert 0 < diff[wandered].max() < 1e-8 and (diff[wandered] > 0).mean() > 0.01   # ... whose wandering end points carry the exact values
