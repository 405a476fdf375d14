"""CPU: the N > 1 path — contiguous live-point shards + one all-gather — with torch.distributed/gloo,
world_size 2 (and 3, ragged).  The evaluator injected here is the oracle (this is a test); the product
passes GpuRVModel and the RCCL transport."""
import os
import socket

import numpy as np
import pytest

from evidence_amd.sharded import ShardedLogLike, padded_count, partition, unpad


def test_partition_and_unpad():
    assert partition(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert partition(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)] and partition(0, 2) == [(0, 0), (0, 0)]
    for n, w in [(10, 4), (16384, 8), (5, 2), (3, 4), (1, 1)]:
        pad = padded_count(n, w)
        g = np.full(w * pad, np.nan)
        for r, (lo, hi) in enumerate(partition(n, w)):
            g[r * pad: r * pad + hi - lo] = np.arange(lo, hi)
        assert np.array_equal(unpad(g, n, w), np.arange(n))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import golden
        from oracle.oracle import OracleModel
        case = golden.config_case(3)
        om = OracleModel(case.layout, case.table)
        theta = np.tile(case.theta, (3, 1))[:n]
        calls = []
        def evaluate(x):
            calls.append(len(x))
            return om.loglike(x)
        out = ShardedLogLike(rank, world, evaluate=evaluate, transport="dist")(theta)
        q.put((rank, out, calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 96), (2, 77), (3, 100)])
def test_sharded_loglike_equals_serial(world, n):
    import torch.multiprocessing as mp
    import golden
    from oracle.oracle import OracleModel
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    case = golden.config_case(3)
    serial = OracleModel(case.layout, case.table).loglike(np.tile(case.theta, (3, 1))[:n])
    for rank, out, calls in results:
        assert np.array_equal(out, serial)                       # every rank holds every log-L, in row order
        lo, hi = partition(n, world)[rank]
        assert calls == [hi - lo]                                # and evaluated only its own shard


def _cube_worker(rank, world, port, n, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from evidence_amd.sharded import ShardedPriorLogLike
        cubes = np.random.default_rng(7).random((n, 5))
        calls = []
        def evaluate(c):                                         # stands in for GpuRVModel.prior_loglike_batch
            calls.append(len(c))
            theta = 10.0 * c - 3.0
            return theta, -0.5 * (theta ** 2).sum(axis=1)
        theta, logl = ShardedPriorLogLike(rank, world, evaluate=evaluate, transport="dist")(cubes)
        q.put((rank, theta, logl, calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 64), (2, 33), (3, 50)])
def test_sharded_prior_loglike_returns_theta_and_logl_everywhere(world, n):
    """The cube form: every rank transforms + evaluates its shard, then theta rows AND log-L are gathered."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cube_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cubes = np.random.default_rng(7).random((n, 5))
    theta = 10.0 * cubes - 3.0
    logl = -0.5 * (theta ** 2).sum(axis=1)
    for rank, th, ll, calls in results:
        assert np.array_equal(th, theta) and np.array_equal(ll, logl)
        assert calls == [padded_count(n, world)]                 # its own shard, padded to the common count


def _fake_walk(cube, theta, logl, lstar, chol, wrapped, nsteps, max_rounds, seed, walker_base=0):
    """A deterministic stand-in for GpuRVModel.slice_walk (the GPU walk is tested in tests/test_gpu_walk.py): like
    the kernel, its "random" displacement is a counter-based function of (seed, walker_base + row)."""
    ids = (np.arange(len(cube), dtype=np.uint64) + np.uint64(walker_base) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
    z = (ids + np.uint64(seed)) ^ ((ids + np.uint64(seed)) >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    c = (cube * 0.5 + u[:, None]) % 1.0
    return c, 10.0 * c - 3.0, logl + u, 3 * len(cube)


def _walk_worker(rank, world, port, n, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from evidence_amd.sharded import ShardedWalker
        rng = np.random.default_rng(3)
        cube, logl = rng.random((n, 4)), rng.normal(size=n)
        calls = []
        def walk(*a, **kw):
            calls.append((len(a[0]), kw.get("walker_base")))
            return _fake_walk(*a, **kw)
        out = ShardedWalker(rank, world, walk, transport="dist")(cube, 10.0 * cube - 3.0, logl, -1.0, np.eye(4), None, 5, 200, 17)
        q.put((rank, out, calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 40), (3, 31), (2, 1)])
def test_sharded_walker_gathers_every_walkers_end_point(world, n):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_walk_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(3)
    cube, logl = rng.random((n, 4)), rng.normal(size=n)
    c, t, l, used = _fake_walk(cube, None, logl, 0, 0, 0, 0, 0, 17)            # the UNSHARDED walk, same seed
    for rank, (gc, gt, gl, gused), calls in results:
        # every rank holds every walker's end point, and it is what the unsharded walk gives: the result depends on
        # the seed only, not on the rank count (each shard passes the common seed and its first row as walker_base)
        assert np.array_equal(gc, c) and np.array_equal(gt, t) and np.array_equal(gl, l) and gused == 3 * n
        lo, hi = partition(n, world)[rank]
        assert calls == ([(hi - lo, lo)] if hi > lo else [])


def test_shard_random_streams_do_not_overlap():
    """per-shard seeds of the form seed + G*(rank+1), with G the counter increment of the generator,
    made rank b replay rank a's stream a few draws later.  With a common seed and walker_base = first row the
    (walker, move, draw) counters of different shards are disjoint by construction."""
    n, world = 1000, 4
    seen = set()
    for lo, hi in partition(n, world):
        ids = set(range(lo, hi))                      # walker_base + row for the rows of this shard
        assert not (ids & seen)
        seen |= ids
    assert seen == set(range(n))
    a = _fake_walk(np.zeros((5, 2)), None, np.zeros(5), 0, 0, 0, 0, 0, 9, walker_base=0)[0]
    b = _fake_walk(np.zeros((5, 2)), None, np.zeros(5), 0, 0, 0, 0, 0, 9, walker_base=5)[0]
    assert not np.intersect1d(a[:, 0], b[:, 0]).size  # different rows, different draws


def test_default_transport_is_rccl_or_the_rendezvous_never_torch():
    """the sharded wrappers defaulted to transport="dist", which imports torch — and a process that
    imports torch before its first HIP call binds torch's bundled HIP runtime and RCCL.  The product default is RCCL
    through the model's communicator (or the rendezvous sockets when only a group is given); gloo is opt-in."""
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parents[1]
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from evidence_amd.sharded import ShardedLogLike, ShardedPriorLogLike, ShardedWalker\n"
        "from evidence_amd.rendezvous import Rendezvous\n"
        "class M:\n"
        "    def log_likelihood_batch(self, x): return -(x ** 2).sum(axis=1)\n"
        "    def prior_loglike_batch(self, c): return c, -(c ** 2).sum(axis=1)\n"
        "    def slice_walk(self, *a, **k): raise AssertionError\n"
        "m, rz = M(), Rendezvous(0, 1)\n"
        "assert ShardedLogLike(0, 1, model=m).transport == 'rccl'\n"
        "assert ShardedPriorLogLike(0, 1, model=m).transport == 'rccl'\n"
        "assert ShardedWalker(0, 1, m.slice_walk, model=m).transport == 'rccl'\n"
        "s = ShardedLogLike(0, 1, evaluate=m.log_likelihood_batch, group=rz)\n"
        "assert s.transport == 'rdzv' and s(np.ones((5, 2))).shape == (5,)\n"
        "assert ShardedWalker(0, 1, m.slice_walk, group=rz).transport == 'rdzv'\n"
        "for make in (lambda: ShardedLogLike(0, 1, evaluate=m.log_likelihood_batch), lambda: ShardedWalker(0, 1, m.slice_walk)):\n"
        "    try: make()\n"
        "    except ValueError as e: assert 'opt-in' in str(e)\n"
        "    else: raise AssertionError('no transport chosen silently')\n"
        "print('torch' in sys.modules)\n") % str(repo)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "False"
