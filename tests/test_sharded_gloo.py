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


def _fake_walk(cube, theta, logl, lstar, chol, wrapped, nsteps, max_rounds, seed):
    """A deterministic stand-in for GpuRVModel.slice_walk (the GPU walk is tested in tests/test_gpu_walk.py)."""
    c = (cube * 0.5 + 0.25) % 1.0
    return c, 10.0 * c - 3.0, logl + 1.0, 3 * len(cube)


def _walk_worker(rank, world, port, n, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from evidence_amd.sharded import ShardedWalker
        rng = np.random.default_rng(3)
        cube, logl = rng.random((n, 4)), rng.normal(size=n)
        calls = []
        def walk(*a):
            calls.append(len(a[0]))
            return _fake_walk(*a)
        out = ShardedWalker(rank, world, walk)(cube, 10.0 * cube - 3.0, logl, -1.0, np.eye(4), None, 5, 200, 17)
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
    c, t, l, used = _fake_walk(cube, None, logl, 0, 0, 0, 0, 0, 0)
    for rank, (gc, gt, gl, gused), calls in results:
        assert np.array_equal(gc, c) and np.array_equal(gt, t) and np.array_equal(gl, l) and gused == 3 * n
        lo, hi = partition(n, world)[rank]
        assert calls == ([hi - lo] if hi > lo else [])
