"""Multi-GPU: live points shard across ranks (one process per GPU); one all-gather of the
per-shard log-L returns every value to every rank — in particular to rank 0, which owns the
sampler's replacement step.  This replaces the MPI fan-out the reference leaves to its third-party
samplers (evidence/polychord/__init__.py:21-29,176-199; evidence/ultranest/__init__.py:21-29,151-194);
there is no other exchange on this path, so there is no other collective — except that when theta itself was
produced on the device from a cube shard (ShardedPriorLogLike), its rows travel back the same way.

Three transports for the same partition:
  "rccl"  the device buffer the log-L kernel wrote is all-gathered in place by RCCL over xGMI on the
          handle's stream (rvll_allgather_logl; host-side rows: rvll_allgather_host) — the product path on a GPU node,
          and the DEFAULT whenever a model is passed;
  "rdzv"  host buffers over evidence_amd/rendezvous.py (plain sockets, no torch) — the fallback transport if RCCL
          cannot be initialised, what a launcher-less script can use, and the default when only group= is passed;
  "dist"  torch.distributed all_gather of host buffers (gloo) — opt-in (transport="dist"): the CPU tests of the
          sharding logic use it.  It is never chosen by default: importing torch before the first HIP call binds
          torch's bundled HIP runtime and RCCL instead of the ROCm ones librvll.so is built against.
The evaluation itself is always whatever `evaluate` is: GpuRVModel.log_likelihood_batch in the product.
"""
from typing import Callable, List, Tuple

import numpy as np


def partition(n_points: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, near-even row ranges [lo, hi) per rank; the first n_points % world ranks get one more."""
    if world < 1 or n_points < 0:
        raise ValueError("world >= 1 and n_points >= 0 required")
    base, extra = divmod(n_points, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def padded_count(n_points: int, world: int) -> int:
    """All-gather needs equal counts: every rank contributes ceil(n/world) slots."""
    return -(-n_points // world) if n_points else 0


def gather_rows(mine: np.ndarray, world: int, transport: str, model=None, group=None) -> np.ndarray:
    """All-gather of equally shaped float64 host arrays: -> [world, *mine.shape] on every rank."""
    mine = np.ascontiguousarray(mine, dtype=np.float64)
    if transport == "rccl":
        return model.allgather_host(mine.ravel(), world).reshape((world,) + mine.shape)
    if transport == "rdzv":
        return np.stack(group.allgather(mine))
    import torch
    import torch.distributed as dist
    parts = [torch.empty(mine.shape, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(parts, torch.from_numpy(mine.copy()))
    return torch.stack(parts).numpy()


def _check_transport(transport, model, group):
    """-> the transport to use.  None picks the product default: "rccl" through the model's communicator when a model
    is given, else "rdzv" over the given rendezvous; torch ("dist") only ever on request."""
    if transport is None:
        if group is not None and model is None:
            transport = "rdzv"
        elif model is not None:
            transport = "rccl"
        else:
            raise ValueError('pass model= (RCCL, the default), group= (rendezvous sockets) or transport="dist" (torch / gloo, opt-in)')
    if transport not in ("dist", "rccl", "rdzv"):
        raise ValueError(transport)
    if transport == "rccl" and model is None:
        raise ValueError("the rccl transport gathers through the model's communicator: pass model=")
    if transport == "rdzv" and group is None:
        raise ValueError("the rdzv transport needs group= (an evidence_amd.rendezvous.Rendezvous)")
    return transport


def unpad(gathered: np.ndarray, n_points: int, world: int) -> np.ndarray:
    """[world * padded] rank-major -> [n_points] in original row order."""
    pad = padded_count(n_points, world)
    g = np.asarray(gathered).reshape(world, pad) if pad else np.empty((world, 0))
    return np.concatenate([g[r, : hi - lo] for r, (lo, hi) in enumerate(partition(n_points, world))]) \
        if n_points else np.empty(0)


class ShardedLogLike:
    """loglike over a batch every rank holds (replicated theta): each rank evaluates its rows, then
    one all-gather.  Returns the full [n] log-L vector on every rank."""

    def __init__(self, rank: int, world: int, evaluate: Callable[[np.ndarray], np.ndarray] = None,
                 model=None, transport: str = None, group=None):
        transport = _check_transport(transport, model, group)
        if evaluate is None and model is None:
            raise ValueError("pass evaluate= or model=")
        self.rank, self.world, self.model, self.transport, self.group = rank, world, model, transport, group
        self.evaluate = evaluate if evaluate is not None else model.log_likelihood_batch

    def __call__(self, theta: np.ndarray) -> np.ndarray:
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        n = theta.shape[0]
        lo, hi = partition(n, self.world)[self.rank]
        pad = padded_count(n, self.world)
        if n == 0:
            return np.empty(0)
        shard = theta[lo:hi]
        if self.transport == "rccl":
            m = self.model
            rows = np.zeros((pad, theta.shape[1]))
            rows[: hi - lo] = shard
            if hi - lo < pad:                      # padding rows: repeat a real row so the kernel sees sane input
                rows[hi - lo:] = theta[lo] if hi > lo else theta[0]
            m.dev_upload_theta(rows)
            m.dev_loglike(pad)
            m.allgather_logl(pad)
            return unpad(m.download_gathered(self.world * pad), n, self.world)
        mine = np.zeros(pad)
        if hi > lo:
            mine[: hi - lo] = np.asarray(self.evaluate(shard), dtype=np.float64)
        return unpad(gather_rows(mine, self.world, self.transport, group=self.group).ravel(), n, self.world)


class ShardedPriorLogLike:
    """prior(cube) + loglike(theta) over a batch of unit-cube rows every rank holds: each rank transforms and
    evaluates its rows on its GPU, then TWO all-gathers — log-L, and the theta rows the prior kernel produced
    on the device (the sampler on rank 0 needs the physical parameters of the points it keeps).
    Returns (theta [n, ndim], logL [n]) on every rank.  Transports as in ShardedLogLike; with "dist" the
    evaluation is `evaluate(cubes) -> (theta, logL)` (GpuRVModel.prior_loglike_batch in the product)."""

    def __init__(self, rank: int, world: int, evaluate: Callable = None, model=None, transport: str = None,
                 group=None):
        transport = _check_transport(transport, model, group)
        if evaluate is None and model is None:
            raise ValueError("pass evaluate= or model=")
        self.rank, self.world, self.model, self.transport, self.group = rank, world, model, transport, group
        self.evaluate = evaluate if evaluate is not None else model.prior_loglike_batch

    def __call__(self, cubes: np.ndarray):
        cubes = np.ascontiguousarray(cubes, dtype=np.float64)
        n, ndim = cubes.shape
        if n == 0:
            return np.empty((0, ndim)), np.empty(0)
        lo, hi = partition(n, self.world)[self.rank]
        pad = padded_count(n, self.world)
        rows = np.full((pad, ndim), 0.5)                     # padding rows: the middle of the cube is always valid
        rows[: hi - lo] = cubes[lo:hi]
        if self.transport == "rccl":
            m = self.model
            m.dev_upload_cube(rows)
            m.dev_prior_loglike(pad)
            m.allgather_theta(pad)
            m.allgather_logl(pad)
            theta_all = m.download_gathered_theta(self.world * pad)
            logl_all = m.download_gathered(self.world * pad)
        else:
            th, ll = self.evaluate(rows)
            mine = np.concatenate([np.asarray(th, dtype=np.float64).reshape(pad, ndim),
                                   np.asarray(ll, dtype=np.float64).reshape(pad, 1)], axis=1)
            both = gather_rows(mine, self.world, self.transport, group=self.group).reshape(self.world * pad, ndim + 1)
            theta_all, logl_all = both[:, :ndim], both[:, ndim]
        keep = np.concatenate([np.arange(r * pad, r * pad + (h - l)) for r, (l, h) in enumerate(partition(n, self.world))])
        return np.ascontiguousarray(theta_all[keep]), np.ascontiguousarray(logl_all[keep])


class ShardedWalker:
    """The sampler's proposal walk (GpuRVModel.slice_walk) sharded over ranks: every rank runs the same sampler
    state (same seed), walks its contiguous share of the replacement walkers on its GPU, and one all-gather
    returns every walker's end point (cube, theta, log-L) and the call count to every rank.  Drop-in for the
    `walker=` argument of nested.run_nested_slice.

    Every rank passes the COMMON seed and the index of its first row as `walker_base`: the walk's random numbers are
    counter-based on (seed, walker_base + row, move, draw), so each walker draws exactly what it would in an
    unsharded walk — a run is reproducible for a given seed whatever the rank count, and no two shards share a
    random stream (deriving per-rank seeds by adding multiples of the counter increment made rank b replay rank
    a's stream a few draws later).

    walk(cube, theta, logl, lstar, chol, wrapped, nsteps, max_rounds, seed, walker_base=...) is
    GpuRVModel.slice_walk in the product.  Transports as in ShardedLogLike ("rccl": model=, through
    rvll_allgather_host)."""

    def __init__(self, rank: int, world: int, walk: Callable, transport: str = None, model=None, group=None):
        transport = _check_transport(transport, model, group)
        self.rank, self.world, self.walk = rank, world, walk
        self.transport, self.model, self.group = transport, model, group

    def __call__(self, cube, theta, logl, lstar, chol, wrapped, nsteps, max_rounds, seed):
        cube = np.ascontiguousarray(cube, dtype=np.float64)
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        logl = np.ascontiguousarray(logl, dtype=np.float64)
        n, ndim = cube.shape
        lo, hi = partition(n, self.world)[self.rank]
        pad = padded_count(n, self.world)
        mine = np.zeros((pad, 2 * ndim + 2))
        if hi > lo:
            c, t, l, used = self.walk(cube[lo:hi], theta[lo:hi], logl[lo:hi], lstar, chol, wrapped, nsteps,
                                      max_rounds, int(seed), walker_base=lo)
            mine[: hi - lo, :ndim], mine[: hi - lo, ndim:2 * ndim], mine[: hi - lo, 2 * ndim] = c, t, l
            mine[0, 2 * ndim + 1] = used
        if pad == 0:
            return cube, theta, logl, 0
        both = gather_rows(mine, self.world, self.transport, model=self.model, group=self.group)
        keep = [both[r, : h - l] for r, (l, h) in enumerate(partition(n, self.world))]
        out = np.concatenate(keep) if keep else np.empty((0, 2 * ndim + 2))
        used_total = int(round(float(both[:, 0, 2 * ndim + 1].sum())))
        return (np.ascontiguousarray(out[:, :ndim]), np.ascontiguousarray(out[:, ndim:2 * ndim]),
                np.ascontiguousarray(out[:, 2 * ndim]), used_total)


class MultiDeviceLogLike:
    """One process, several GPUs: the same contiguous live-point shards, one GpuRVModel (handle) per device,
    one host thread per device (the ctypes calls release the GIL, so uploads, kernels and downloads of the
    devices overlap).  For samplers that run in a single process (UltraNest without MPI, nested.py): no
    launcher, no communicator — the per-shard log-L come back through each device's own PCIe link and are
    concatenated on the host.  The process-per-GPU + RCCL form above is what bench.py scales with."""

    def __init__(self, models):
        from concurrent.futures import ThreadPoolExecutor
        if not models:
            raise ValueError("at least one model")
        self.models = list(models)
        self._pool = ThreadPoolExecutor(max_workers=len(self.models))

    @classmethod
    def create(cls, fixedpardict, datadict, parnames, devices, **kwargs):
        from .engine import GpuRVModel
        return cls([GpuRVModel(fixedpardict, datadict, parnames, device=d, **kwargs) for d in devices])

    @property
    def parnames(self):
        return self.models[0].parnames

    def log_likelihood_batch(self, theta):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        bounds = partition(theta.shape[0], len(self.models))
        futures = [self._pool.submit(m.log_likelihood_batch, theta[lo:hi]) if hi > lo else None
                   for m, (lo, hi) in zip(self.models, bounds)]
        parts = [f.result() if f is not None else np.empty(0) for f in futures]
        return np.concatenate(parts) if parts else np.empty(0)

    def prior_loglike_batch(self, cubes):
        cubes = np.ascontiguousarray(cubes, dtype=np.float64)
        bounds = partition(cubes.shape[0], len(self.models))
        futures = [self._pool.submit(m.prior_loglike_batch, cubes[lo:hi]) if hi > lo else None
                   for m, (lo, hi) in zip(self.models, bounds)]
        res = [f.result() for f in futures if f is not None]
        return np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res])

    def slice_walk(self, cube, theta, logl, lstar, chol, wrapped=None, nsteps=10, max_rounds=200, seed=0):
        """The proposal walk (GpuRVModel.slice_walk) with the walkers sharded over the devices; drop-in for the
        `walker=` argument of nested.run_nested_slice.  Every device gets the same seed and its first row as
        walker_base, so the result does not depend on the number of devices (see ShardedWalker)."""
        cube = np.ascontiguousarray(cube, dtype=np.float64)
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        logl = np.ascontiguousarray(logl, dtype=np.float64)
        bounds = partition(cube.shape[0], len(self.models))
        futures = [self._pool.submit(m.slice_walk, cube[lo:hi], theta[lo:hi], logl[lo:hi], lstar, chol, wrapped, nsteps,
                                     max_rounds, int(seed), lo) if hi > lo else None
                   for m, (lo, hi) in zip(self.models, bounds)]
        res = [f.result() for f in futures if f is not None]
        if not res:
            return cube, theta, logl, 0
        return (np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res]),
                np.concatenate([r[2] for r in res]), int(sum(r[3] for r in res)))

    def close(self):
        self._pool.shutdown(wait=True)
        for m in self.models:
            m.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
