"""GPU: the drop-in boundary — sampler callbacks with the reference's signatures, BASELINE.json
configs[0] through the nested-sampling driver, and the RCCL all-gather on a single rank."""
import numpy as np
import pytest

import golden
from evidence_amd import GpuRVModel, RvllError
from evidence_amd.callbacks import make_polychord_callbacks, make_ultranest_callbacks, wrapped_params
from evidence_amd.nested import run_nested
from evidence_amd.sharded import ShardedLogLike
from evidence_amd.synthetic import make_workload

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180)]


@pytest.fixture(scope="module")
def cfg1(gpu_required):
    w = make_workload(1)
    m = GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict())
    yield w, m
    m.close()


def test_polychord_callback_conventions(cfg1):
    w, m = cfg1
    prior, loglike, ndim, nderived = make_polychord_callbacks(m)
    assert (ndim, nderived) == (6, 0)
    cube = np.full(ndim, 0.37)
    theta = prior(cube)
    assert isinstance(theta, np.ndarray) and theta.shape == (ndim,) and theta.dtype == np.float64
    assert theta is not cube and np.all(cube == 0.37)                    # a NEW array (polychord:138 ones_like)
    out = loglike(theta)
    assert isinstance(out, tuple) and isinstance(out[0], float) and out[1] == []    # (logL, derived) polychord:171


def test_ultranest_callback_conventions_scalar_and_vectorized(cfg1):
    w, m = cfg1
    prior, loglike = make_ultranest_callbacks(m)
    theta = prior(np.full(6, 0.5))
    assert theta.shape == (6,) and isinstance(loglike(theta), float)                 # ultranest:146
    vprior, vloglike = make_ultranest_callbacks(m, vectorized=True)
    cubes = w.sample_cube(100, 3)
    thetas = vprior(cubes)
    ll = vloglike(thetas)
    assert thetas.shape == (100, 6) and ll.shape == (100,)
    assert ll[0] == loglike(prior(cubes[0]))                                         # scalar == row of the batch
    assert list(wrapped_params(m.parnames)) == [("omega" in p or "ml0" in p) for p in m.parnames]


def test_config0_400_live_points_through_the_callbacks(cfg1):
    """BASELINE.json configs[0]: 1-planet circular, 50 epochs, 400 live points via the (UltraNest-style)
    callbacks.  The same run with the oracle as likelihood must take the same path: same ln Z.
    Only the first 1500 replacements are run: the period posterior of an RV signal is multimodal, which the
    driver's single-ellipsoid rejection sampling is not built for (evidence_amd/nested.py docstring); this test
    is about the callback plumbing, and `max_calls` bounds it regardless."""
    from oracle.oracle import OracleModel
    w, m = cfg1
    vprior, vloglike = make_ultranest_callbacks(m, vectorized=True)
    gpu = run_nested(vprior, vloglike, m.ndim, nlive=400, dlogz=0.5, seed=4, max_iter=1500, max_calls=600_000)
    om = OracleModel(m.layout, w.table)
    cpu = run_nested(vprior, lambda t: om.loglike(t, nthreads=8), m.ndim, nlive=400, dlogz=0.5, seed=4, max_iter=1500,
                     max_calls=600_000)
    assert gpu.niter == cpu.niter and gpu.ncall == cpu.ncall
    assert abs(gpu.logz - cpu.logz) <= 1e-9 * abs(cpu.logz)
    assert np.isfinite(gpu.logz) and gpu.niter >= 400


def test_gaussian_known_answer_is_reproduced_with_device_priors(gpu_required):
    """The reference's 1-D/2-D Gaussian tests use Uniform(-10,10) priors: take the prior transform from the
    GPU (two offset parameters) and the toy likelihood from numpy, as tests/test_polychord.py does."""
    from evidence_amd import priors as P
    from evidence_amd.data import EpochTable
    table = EpochTable.from_arrays(["a", "b"], [1.0, 2.0], [0.0, 0.0], [1.0, 1.0], [0, 1])
    pri = {"a_offset": P.Uniform(-10, 10), "b_offset": P.Uniform(-10, 10)}
    with GpuRVModel({}, table, list(pri), priordict=pri) as m:
        res = run_nested(m.prior_transform_batch, lambda x: -0.5 * np.sum(x * x, axis=1), 2, nlive=500, dlogz=0.05, seed=1)
    assert abs(res.logz - (-4.1536)) < 0.5                                           # tests/test_polychord.py:139


def test_rccl_allgather_single_rank_and_sharded_wrapper(gpu_required):
    case = golden.config_case(3)
    with GpuRVModel(case.fixed, case.table, case.parnames) as m:
        m.comm_init(GpuRVModel.comm_unique_id(), 1, 0)
        want = m.log_likelihood_batch(case.theta)
        got = ShardedLogLike(0, 1, model=m, transport="rccl")(case.theta)
        assert np.array_equal(got, want)
        # back-to-back launches alternate the two log-L buffers while gathers are in flight
        m.dev_upload_theta(case.theta)
        for _ in range(5):
            m.dev_loglike(len(case.theta))
            m.allgather_logl(len(case.theta))
        assert np.array_equal(m.download_gathered(len(case.theta)), want)
        m.comm_destroy()


def test_rccl_theta_gather_single_rank_cube_form(gpu_required):
    """The cube form of the multi-GPU step on a one-rank RCCL communicator: theta produced on the device comes
    back through rvll_allgather_theta next to the log-L gather; repeated steps keep the two collectives ordered."""
    from evidence_amd.sharded import ShardedPriorLogLike
    w = make_workload(3)
    cubes = w.sample_cube(1000, seed=21)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        want_theta, want_logl = m.prior_loglike_batch(cubes)
        m.comm_init(GpuRVModel.comm_unique_id(), 1, 0)
        sharded = ShardedPriorLogLike(0, 1, model=m, transport="rccl")
        for _ in range(4):
            theta, logl = sharded(cubes)
            assert np.array_equal(theta, want_theta) and np.array_equal(logl, want_logl)
        m.comm_destroy()
        from evidence_amd import RvllError
        with pytest.raises(RvllError):
            m.allgather_theta(10)                       # no communicator any more


def test_reference_style_config_end_to_end(gpu_required):
    """examples/51peg/config_51peg.py (the reference's config-module format) -> read_config -> GpuRVModel:
    the BASELINE.md known answers for the shipped 51Peg example come out of the whole chain."""
    from pathlib import Path
    from evidence_amd.config import read_config
    cfg = Path(__file__).resolve().parents[1] / "examples" / "51peg" / "config_51peg.py"
    rundict, datadict, priordict, fixed = read_config(cfg, nplanets=1)
    with GpuRVModel(fixed, datadict, list(priordict), priordict=priordict) as m:
        assert m.parnames == ["hamilton_jitter", "hamilton_offset", "planet1_ecc", "planet1_k1", "planet1_ma0",
                              "planet1_omega", "planet1_period"] and m.nplanets == 1
        logl = m.log_likelihood([3.0, -2.0, 0.05, 56.0, 1.0, 0.7, 4.2308])
        theta = m.prior_transform(np.full(7, 0.37))
    assert abs(logl - (-11539.57252446112)) <= 1e-10 * 11539.6
    want = [18.5, -2.5999999999999996, 0.11465599401597197, 1.288249551693134, 2.324778563656447,
            2.324778563656447, 1.5780337699226765]
    assert np.max(np.abs(theta - want) / np.abs(want)) <= 1e-13


def test_51peg_evidence_run_is_gpu_fed_and_reproducible(gpu_required):
    """End-to-end evidence run on the shipped 51 Peg example through the batched slice sampler: the GPU and
    the oracle likelihood take the same path for the same seed, and two seeds agree within their errors."""
    from pathlib import Path
    from evidence_amd.config import read_config
    from evidence_amd.nested import run_nested_slice
    from oracle.oracle import OracleModel
    cfg = Path(__file__).resolve().parents[1] / "examples" / "51peg" / "config_51peg.py"
    rundict, datadict, priordict, fixed = read_config(cfg, nplanets=1)
    with GpuRVModel(fixed, datadict, list(priordict), priordict=priordict) as m:
        vprior, vloglike = make_ultranest_callbacks(m, vectorized=True)
        wrap = wrapped_params(m.parnames)
        kw = dict(nlive=200, dlogz=0.5, wrapped=wrap, max_calls=3_000_000)
        a = run_nested_slice(vprior, vloglike, m.ndim, seed=1, **kw)
        b = run_nested_slice(vprior, vloglike, m.ndim, seed=2, **kw)
        om = OracleModel(m.layout, m.table)
        short = dict(nlive=100, dlogz=0.5, wrapped=wrap, max_iter=400)
        g = run_nested_slice(vprior, vloglike, m.ndim, seed=3, **short)
        c = run_nested_slice(vprior, lambda t: om.loglike(t, nthreads=8), m.ndim, seed=3, **short)
    assert g.ncall == c.ncall and abs(g.logz - c.logz) <= 1e-9 * abs(c.logz)
    # seed-to-seed scatter of ln Z on this multimodal posterior is ~1.5 at this live-point count
    assert abs(a.logz - b.logz) < 5 * np.hypot(a.logzerr, b.logzerr) + 3.0, (a.logz, b.logz, a.logzerr, b.logzerr)
    # the posterior finds the planet: P = 4.2308 d, K ~ 56 m/s (the known 51 Peg b)
    w = np.exp(a.logwt)
    ip, ik = m.parnames.index("planet1_period"), m.parnames.index("planet1_k1")
    assert abs(np.sum(w * a.samples[:, ip]) - 4.2308) < 0.01
    assert abs(np.sum(w * a.samples[:, ik]) - 56.0) < 6.0


def test_single_process_multi_handle_sharding(gpu_required):
    """MultiDeviceLogLike with two handles (both on device 0 here; on a node: one per GPU): same values, same
    order as one handle, for even and ragged batch sizes and for the fused prior+log-L call."""
    from evidence_amd.sharded import MultiDeviceLogLike
    w = make_workload(3)
    pri = w.priordict()
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=pri) as one, \
            MultiDeviceLogLike.create(w.fixedpardict, w.table, w.parnames, devices=[0, 0, 0], priordict=pri) as multi:
        for n in (3000, 3001, 2, 1):
            theta = w.sample_theta(n, seed=n)
            assert np.array_equal(multi.log_likelihood_batch(theta), one.log_likelihood_batch(theta))
        cube = w.sample_cube(1000, 5)
        t1, l1 = one.prior_loglike_batch(cube)
        t2, l2 = multi.prior_loglike_batch(cube)
        assert np.array_equal(t1, t2) and np.array_equal(l1, l2)


def test_single_process_multi_handle_walk(gpu_required):
    """MultiDeviceLogLike.slice_walk: walkers sharded over two handles (one per GPU on a node); every end point
    is above the threshold and consistent with a fresh evaluation, whatever handle walked it."""
    from evidence_amd.sharded import MultiDeviceLogLike
    w = make_workload(3)
    with MultiDeviceLogLike.create(w.fixedpardict, w.table, w.parnames, devices=[0, 0], priordict=w.priordict()) as md:
        cube = np.random.default_rng(2).random((3001, len(w.parnames)))
        theta, logl = md.prior_loglike_batch(cube)
        lstar = float(np.median(logl))
        keep = logl > lstar
        cube, theta, logl = cube[keep], theta[keep], logl[keep]
        c2, t2, l2, n = md.slice_walk(cube, theta, logl, lstar, 0.1 * np.eye(cube.shape[1]),
                                      wrapped_params(w.parnames), nsteps=8, seed=4)
        th_chk, ll_chk = md.prior_loglike_batch(c2)
    assert c2.shape == cube.shape and n >= 8 * len(cube)
    assert (l2 > lstar).all() and np.array_equal(th_chk, t2) and np.array_equal(ll_chk, l2)


def test_pipeline_lanes_give_identical_results_and_order_theta_updates(gpu_required):
    """Device-resident launches alternate between two lanes (streams + buffers).  Either lane must produce the
    same log-L, and a theta upload / prior launch (lane 0's stream) must be seen by a following lane-1 launch."""
    w = make_workload(3)
    t1, t2 = w.sample_theta(3000, seed=21), w.sample_theta(3000, seed=22)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        want1, want2 = m.log_likelihood_batch(t1), m.log_likelihood_batch(t2)
        m.dev_upload_theta(t1)
        m.dev_loglike(3000)                                   # lane 0
        assert m.dev_flip_lane() == 1
        m.dev_upload_theta(t2)                                # async copy on lane 0's stream ...
        m.dev_loglike(3000)                                   # ... must be visible to this lane-1 launch
        assert np.array_equal(m.dev_download(3000)[1], want2)
        assert m.dev_flip_lane() == 0
        m.dev_upload_theta(t1)                                # waits for lane 1's readers before overwriting theta
        m.dev_loglike(3000)
        assert np.array_equal(m.dev_download(3000)[1], want1)
        for _ in range(7):                                    # many launches in flight on alternating lanes
            m.dev_loglike(3000)
            m.dev_flip_lane()
        m.dev_sync()
        if m.dev_flip_lane() != 0:
            m.dev_flip_lane()
        cube = w.sample_cube(3000, 1)
        m.dev_upload_cube(cube)
        m.dev_prior(3000)
        m.dev_flip_lane()
        m.dev_loglike(3000)                                   # lane 1 behind the prior kernel on lane 0
        theta, logl, _ = m.dev_download(3000, theta=True)
        assert np.array_equal(logl, m.log_likelihood_batch(theta))


def test_pipeline_lanes_are_added_collectively_and_every_lane_gathers(gpu_required):
    """rvll_comm_init gives one lane (the gather runs in-stream behind its kernel); rvll_comm_add_lanes splits further
    communicators and rvll_comm_set_lanes applies the count the ranks agreed on (here: one rank).  Steps then cycle
    through the lanes — each with its own stream, communicator and buffers — and every lane's gather returns the
    values of the kernel that ran on it; going back to one lane works too."""
    case = golden.config_case(3)
    n = len(case.theta)
    with GpuRVModel(case.fixed, case.table, case.parnames) as m:
        want = m.log_likelihood_batch(case.theta)
        m.comm_init(GpuRVModel.comm_unique_id(), 1, 0)
        m.dev_upload_theta(case.theta)
        for _ in range(3):                                   # one lane
            m.dev_loglike(n)
            m.allgather_logl(n)
        assert np.array_equal(m.download_gathered(n), want)
        have = m.comm_add_lanes(3)
        assert 1 <= have <= 3
        m.comm_set_lanes(have)
        for k in range(2 * have + 1):                        # every lane, more than once
            m.dev_loglike(n)
            m.allgather_logl(n)
            assert np.array_equal(m.download_gathered(n), want), k
        with pytest.raises(RvllError):
            m.comm_set_lanes(have + 1 if have < 4 else 5)    # more lanes than communicators
        m.comm_set_lanes(1)
        m.dev_loglike(n)
        m.allgather_logl(n)
        assert np.array_equal(m.download_gathered(n), want)
        info = GpuRVModel.runtime_info()
        assert info["librccl"].endswith(("librccl.so.1", "librccl.so")) and info["rccl_version"] > 0
        assert info["libamdhip64"].rsplit("/", 1)[0] == info["librccl"].rsplit("/", 1)[0]     # the pair from one ROCm
        m.comm_destroy()


def test_host_rows_over_rccl_and_the_sharded_walker(gpu_required):
    """rvll_allgather_host (one rank): the sampler's sharded host state goes through the device and RCCL and comes
    back; ShardedWalker on that transport returns what the plain walk returns."""
    from evidence_amd.callbacks import wrapped_params
    from evidence_amd.sharded import ShardedWalker
    w = make_workload(3)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        m.comm_init(GpuRVModel.comm_unique_id(), 1, 0)
        x = np.random.default_rng(0).normal(size=(37, 5))
        assert np.array_equal(m.allgather_host(x, 1).reshape(x.shape), x)
        cube = np.random.default_rng(1).random((500, m.ndim))
        theta, logl = m.prior_loglike_batch(cube)
        lstar = float(np.median(logl))
        keep = logl > lstar
        chol = np.eye(m.ndim) * 0.1
        wr = wrapped_params(m.parnames)
        plain = m.slice_walk(cube[keep], theta[keep], logl[keep], lstar, chol, wr, nsteps=5, seed=3)
        sharded = ShardedWalker(0, 1, m.slice_walk, transport="rccl", model=m)(
            cube[keep], theta[keep], logl[keep], lstar, chol, wr, 5, 200, 3)
        for a, b in zip(plain, sharded):
            assert np.array_equal(a, b)
        m.comm_destroy()


@pytest.mark.parametrize("n", [16384, 40001, 131072])
def test_large_host_batches_go_up_in_overlapped_chunks(gpu_required, n, monkeypatch):
    """From 16384 points on, rvll_loglike_batch uploads and evaluates a host batch in chunks on two streams
    (upload of chunk i+1 overlaps the kernel of chunk i): same bits as the single-shot path, ragged sizes too."""
    from evidence_amd.synthetic import make_workload
    w = make_workload(3)
    theta = w.sample_theta(n, seed=n)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
        got, flags = m.log_likelihood_batch(theta, return_flags=True)
        monkeypatch.setenv("RVLL_SPLIT", "1")
        whole, flags1 = m.log_likelihood_batch(theta, return_flags=True)
        monkeypatch.setenv("RVLL_SPLIT", "7")
        seven = m.log_likelihood_batch(theta)
    assert np.array_equal(got, whole) and np.array_equal(flags, flags1) and np.array_equal(seven, whole)


@pytest.mark.parametrize("chunk,workers,lag", [(4096, 3, 2), (16384, 8, 2), (30000, 1, 1)])
def test_streamed_host_batches_are_the_device_resident_bits(gpu_required, monkeypatch, chunk, workers, lag):
    """Host batches of 24 MB of rows and more go through pinned staging blocks in chunks, the host's copies on worker threads
    (csrc/rvll_api.hip, stream_host_batch): the same kernels on the same rows.  Forced here for every size (ragged last
    chunk, fewer rows than a chunk, fewer chunks than the pipeline is deep), into fresh, recycled and caller-owned arrays."""
    n = 230_001                                                           # 35 MB of rows: a recycled result block (engine.py)
    w = make_workload(3)
    cube = w.sample_cube(n, seed=5)
    cube[::1013] *= 1e-9
    monkeypatch.setenv("RVLL_COPY_THREADS", str(workers))
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        m.dev_upload_cube(cube); m.dev_prior(n); m.dev_loglike(n)
        theta, logl, flags = (a.copy() for a in m.dev_download(n, theta=True, flags=True))
        monkeypatch.setenv("RVLL_STREAM_MIN", "1")
        monkeypatch.setenv("RVLL_STREAM_CHUNK", str(chunk))
        monkeypatch.setenv("RVLL_STREAM_LAG", str(lag))
        for k in (n, 3, chunk - 1, chunk, chunk + 1, 2 * chunk + 7, 70_000):
            for rep in range(2):                                          # (the second call of the full size reuses the first one's block)
                th, ll, fl = m.prior_loglike_batch(cube[:k], return_flags=True)
                assert np.array_equal(th, theta[:k]) and np.array_equal(ll, logl[:k]) and np.array_equal(fl, flags[:k]), k
                del th
            ll, fl = m.log_likelihood_batch(theta[:k], return_flags=True)
            assert np.array_equal(ll, logl[:k]) and np.array_equal(fl, flags[:k]), k
        mine_t, mine_l = np.full((n, m.ndim), np.nan), np.full(n, np.nan)
        th, ll = m.prior_loglike_batch(cube, theta_out=mine_t, logl_out=mine_l)
        assert th is mine_t and ll is mine_l and np.array_equal(mine_t, theta) and np.array_equal(mine_l, logl)
        with pytest.raises(ValueError, match="theta_out"):
            m.prior_loglike_batch(cube, theta_out=mine_t[:-1])
        with pytest.raises(ValueError, match="logl_out"):
            m.prior_loglike_batch(cube, logl_out=mine_l.astype(np.float32))
        # and the routes it replaced, on the same handle afterwards
        monkeypatch.setenv("RVLL_STREAM_MIN", str(1 << 40))
        th, ll = m.prior_loglike_batch(cube[:70_000])
        assert np.array_equal(th, theta[:70_000]) and np.array_equal(ll, logl[:70_000])


def test_scalar_server_answers_polychord_style_calls(gpu_required):
    """The persistent scalar-call kernel (rvll_scalar_server): same bits as the launch-per-call path, flags
    included; survives interleaved batch / prior calls (each stops it), its own idle exit, a second model with
    its own server, and destruction while running."""
    import time
    case = golden.config_case(3)
    bad = case.theta[5].copy()
    bad[case.parnames.index("planet1_period")] = np.nan
    with GpuRVModel(case.fixed, case.table, case.parnames) as m, \
         GpuRVModel(case.fixed, case.table, case.parnames) as m2:
        want = m.log_likelihood_batch(case.theta)
        m.scalar_server(True)
        m2.scalar_server(True)
        got = np.array([m.log_likelihood(x) for x in case.theta[:64]])
        assert np.array_equal(got, want[:64])
        assert np.array_equal(np.array([m2.log_likelihood(x) for x in case.theta[64:96]]), want[64:96])
        # a batch call in between stops the server; the next scalar call starts it again
        assert np.array_equal(m.log_likelihood_batch(case.theta[:300]), want[:300])
        assert m.log_likelihood(case.theta[7]) == want[7]
        one, flag = m.log_likelihood_batch(case.theta[9:10], return_flags=True)
        assert one[0] == want[9] and flag[0] == 0
        # idle exit (5 ms without a request), then restart on demand
        time.sleep(0.05)
        assert m.log_likelihood(case.theta[11]) == want[11]
        # non-finite input behaves as in the launch path
        m.scalar_server(False)
        ref_bad = m.log_likelihood(bad)
        m.scalar_server(True)
        got_bad = m.log_likelihood(bad)
        assert (np.isnan(ref_bad) and np.isnan(got_bad)) or ref_bad == got_bad
        # PolyChord-convention closure on top of it
        prior, loglike, ndim, nderived = make_polychord_callbacks(m)
        val, derived = loglike(case.theta[3])
        assert val == want[3] and derived == []
    # prior(cube) and loglike(theta) alternate per point, as PolyChord calls them: both go through the server
    w = make_workload(3)
    cubes = w.sample_cube(40, seed=4)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        th_want, ll_want = m.prior_loglike_batch(cubes)
        m.scalar_server(True)
        prior, loglike, ndim, nderived = make_polychord_callbacks(m)
        for c, th_w, ll_w in zip(cubes, th_want, ll_want):
            th = prior(c)
            assert np.array_equal(th, th_w)
            assert loglike(th)[0] == ll_w
        # the low-latency closures ask for prior + log-L in ONE request and answer loglike(theta) from it — only when
        # handed exactly the theta prior() returned; same bits as the two separate calls, with or without the server
        prior2, loglike2, _, _ = make_polychord_callbacks(m, low_latency=True)
        for c, th_w, ll_w in zip(cubes, th_want, ll_want):
            th = prior2(c)
            assert np.array_equal(th, th_w)
            assert loglike2(th)[0] == ll_w                          # from the pair
            assert loglike2(th_want[0])[0] == ll_want[0]            # any other theta: evaluated as usual
        th1, ll1 = m.prior_loglike(cubes[3])
        m.scalar_server(False)
        th2, ll2 = m.prior_loglike(cubes[3])
        assert np.array_equal(th1, th_want[3]) and np.array_equal(th2, th_want[3]) and ll1 == ll_want[3] and ll2 == ll_want[3]
        # the vectorized pair: transform() evaluates log-L along the way, loglike() answers from it for that very batch
        vprior, vloglike = make_ultranest_callbacks(m, vectorized=True, paired=True)
        th = vprior(cubes)
        assert np.array_equal(th, th_want) and np.array_equal(vloglike(th), ll_want)
        assert np.array_equal(vloglike(th_want[:7]), ll_want[:7])              # another batch: evaluated as usual
        th[0, 0] = th_want[1, 0]                                               # the caller's array is its own
        assert np.array_equal(vloglike(th_want), ll_want)
    # both handles were destroyed with their servers possibly still polling: a new model works at once
    with GpuRVModel(case.fixed, case.table, case.parnames) as m3:
        assert np.array_equal(m3.log_likelihood_batch(case.theta[:10]), want[:10])


def test_every_transport_threshold_one_row_either_side(gpu_required):
    """The three host-buffer entry points choose a transport by the size of the call (scalar server, zero-copy through
    the mapped pinned blocks, pinned landing zone, overlapped chunks, staged downloads: csrc/rvll_api.hip).  Rows are
    independent, so a call on any slice of one table returns the bits of those rows computed once through the
    device-resident path — at every threshold, one row below and one above (scripts/host_call_soak.py soaks the same
    with random sizes; profiles/r02_host_call_soak.txt)."""
    nbig = 166_000
    w = make_workload(3)
    rng = np.random.default_rng(9)
    cube = w.sample_cube(nbig, seed=43)
    cube[::997] *= 1e-9                                                   # rows hard against the cube's walls
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        D = m.ndim
        m.dev_upload_cube(cube); m.dev_prior(nbig); m.dev_loglike(nbig)
        theta, logl, flags = (a.copy() for a in m.dev_download(nbig, theta=True, flags=True))
        sizes = {1, 2, 63, 64, 65, 4095, 4096, 4097, 16383, 16384, 16385, 65535, 65536, 65537, 131071, 131072, 131073, nbig}
        for T in (64 << 10, 384 << 10, 1 << 20, 8 << 20, 24 << 20):       # (the last: where streamed chunks take over)
            for r in (8 * D, 16 * D, 12, 8 * D + 12, 16 * D + 12):
                n = T // r
                sizes.update(k for k in (n - 1, n, n + 1) if 1 <= k <= nbig)
        for n in sorted(sizes):
            o = int(rng.integers(0, nbig - n + 1))
            got, fl = m.log_likelihood_batch(theta[o:o + n], return_flags=True)
            assert np.array_equal(got, logl[o:o + n]) and np.array_equal(fl, flags[o:o + n]), ("loglike", n)
            assert np.array_equal(m.prior_transform_batch(cube[o:o + n]), theta[o:o + n]), ("prior", n)
            th, got, fl = m.prior_loglike_batch(cube[o:o + n], return_flags=True)
            assert np.array_equal(th, theta[o:o + n]) and np.array_equal(got, logl[o:o + n]) and \
                np.array_equal(fl, flags[o:o + n]), ("prior_loglike", n)


def test_scalar_server_with_more_parameters_than_its_polling_wave_has_lanes(gpu_required):
    """Round 4: up to 64 parameters a scalar request and its row arrive in one read (the request word beside every value,
    rvll_kernels.h ServerCtl::in); a larger model keeps the two-step protocol (word, then the row fetched by the tile).  A
    14-planet model (74 free parameters): the server's answers are the batch path's bits either way, the same theta twice in a
    row included (nothing in a slot changes but the request), and a 13-parameter model next to it exercises the one-read path."""
    w = make_workload(3)
    rng = np.random.default_rng(9)
    two_pi = 2 * np.pi

    def model_of(nplanets):
        names = ["harps_jitter", "harps_offset", "hires_jitter", "hires_offset"]
        for n in range(1, nplanets + 1):
            names += [f"planet{n}_{k}" for k in ("ecc", "k1", "ma0", "omega", "period")]
        names = sorted(names)
        fixed = {f"planet{n}_epoch": 51000.0 for n in range(1, nplanets + 1)}
        cols = []
        for name in names:
            kind = name.split("_")[1]
            lo, hi = {"jitter": (0.0, 5.0), "offset": (-5.0, 5.0), "ecc": (0.0, 0.6), "k1": (0.1, 10.0), "ma0": (0.0, two_pi),
                      "omega": (0.0, two_pi), "period": (2.0, 500.0)}[kind]
            cols.append(rng.uniform(lo, hi, 40))
        return fixed, names, np.column_stack(cols)

    for nplanets in (14, 2):
        fixed, names, theta = model_of(nplanets)
        with GpuRVModel(fixed, w.table, names) as m:
            assert m.ndim == 5 * nplanets + 4
            want = m.log_likelihood_batch(theta)
            assert np.isfinite(want).all()
            m.scalar_server(True)
            got = np.array([m.log_likelihood(x) for x in theta])
            again = np.array([m.log_likelihood(theta[7]) for _ in range(5)])
            m.scalar_server(False)
            assert np.array_equal(got, want) and (again == want[7]).all(), nplanets
