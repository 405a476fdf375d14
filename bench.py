#!/usr/bin/env python3
"""bench.py — live-point log-L evaluations per second on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`,
     one rank per GPU; live points shard across ranks, one RCCL all-gather of per-shard log-L per step)

A step = one pass of the hot path over one batch: the fused log-L kernel over B live points
already resident in HBM (theta uploaded before the timed region), followed — for N > 1 — by
the RCCL all-gather of the per-shard log-L to every rank, in-stream behind the kernel; with
pipeline lanes consecutive steps alternate streams / communicators so that the gather of one
step overlaps the kernel of the next.  Weak scaling: B live points PER GPU, fixed as N grows.
No torch anywhere: the ranks meet over evidence_amd/rendezvous.py (a process that imports torch
first binds torch's bundled HIP runtime and RCCL instead of the ROCm ones librvll is built for).

Workload: BASELINE.json configs[2] — 3-planet Keplerian, 200 epochs, 2 instruments with
jitter + offset, 16384 live points (the configuration the >= 1e8 evals/s target is quoted on).

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the field definitions).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from evidence_amd import GpuRVModel  # noqa: E402
from evidence_amd.synthetic import CONFIGS, make_workload  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # vector fp64 = half of the 157.3 TF fp32 vector rate

WORKLOAD_TEXT = {
    1: "cfg1: 1-planet circular, 50 epochs, 1 instrument, {b} live points/GPU",
    2: "cfg2: 1-planet eccentric (e=0.3), 200 epochs, 1 instrument, {b} live points/GPU",
    3: "cfg3: 3-planet Keplerian, 200 epochs, 2 instruments w/ jitter+offset, {b} live points/GPU",
    4: "cfg4: 3-planet Keplerian, 1000 epochs, 2 instruments, {b} live points/GPU",
    5: "cfg5: 5-planet + linear drift, 2000 epochs, 3 instruments, {b} live points/GPU",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="live points per GPU (default: the config's)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the host-roundtrip / prior+loglike extras (profiling)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--points-per-block", type=int, default=0)
    ap.add_argument("--precision", default="fp64", choices=["fp64", "mixed", "fp32"],
                    help="fp64 = the reference's arithmetic (default, the only parity mode)")
    return ap.parse_args()


def algorithmic_bytes_per_launch(ndim, n_epochs, batch):
    """SURVEY.md §8d, L mode: theta in (8 D) + log-L out (8) per live point, + the epoch table
    (t, y, sigma^2 f64 + instrument id i32 = 28 B/epoch) once per launch."""
    return batch * (8 * ndim + 8) + 28 * n_epochs


def flops_per_eval(nplanets, n_epochs, mean_iters):
    """SURVEY.md §8d convention: F_eval = Ne (Np (128 + 96 n_it) + 60)."""
    return n_epochs * (nplanets * (128.0 + 96.0 * mean_iters) + 60.0)


def host_cpu_share():
    """CPU cores this process may actually use: the cgroup quota (the GPU box gives 16 of its 256
    logical CPUs to a one-GPU job), else the affinity mask; RVLL_CPU_THREADS overrides."""
    if os.environ.get("RVLL_CPU_THREADS"):
        return max(1, int(os.environ["RVLL_CPU_THREADS"]))
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(w, layout, theta, gpu_logl, seconds):
    """The oracle (C restatement of the reference path) on this node's host cores, OpenMP over live
    points, on a bounded sample of the same workload.  Reported baseline, not the target.

    SURVEY §8(d) asks for all host cores, and the cgroup's cpu.max understates what a one-GPU job gets on the pool's
    boxes (round 1: 16 by the quota, yet 128 threads ran 6x faster), so the thread count is SWEPT — 1, the cgroup
    share, 64, 128, every logical CPU — for an equal slice of `seconds` each, and the best is the value reported;
    the whole sweep is listed beside it."""
    from oracle import oracle as orc            # checker + CPU baseline only
    om = orc.OracleModel(layout, w.table)
    n = theta.shape[0]
    logical = os.cpu_count() or 1
    if os.environ.get("RVLL_CPU_THREADS"):
        counts = [host_cpu_share()]
    else:
        counts = sorted({c for c in (1, host_cpu_share(), 64, 128, logical) if 1 <= c <= logical})
    ref = om.loglike(theta, nthreads=counts[-1])   # warm + parity sample
    err = np.abs(gpu_logl - ref) / np.maximum(np.abs(ref), 1e-300)
    slice_s = max(0.5, seconds / len(counts))
    swept = {}
    for c in counts:
        sub = theta if c > 1 else theta[: max(1, n // 8)]      # one thread: a slice of the batch per pass
        done, t0 = 0, time.perf_counter()
        while True:
            om.loglike(sub, nthreads=c)
            done += sub.shape[0]
            el = time.perf_counter() - t0
            if el >= slice_s:
                break
        swept[c] = {"evals_per_s": done / el, "evaluations": done, "seconds": el}
    best = max(swept, key=lambda c: swept[c]["evals_per_s"])
    iters = np.concatenate([om.iteration_counts(theta[i]).ravel() for i in range(0, n, max(1, n // 64))])
    b = swept[best]
    return {
        "value": b["evals_per_s"], "unit": "evals/s", "cores": best, "kind": "port",
        "sample": f"{b['evaluations']} evaluations (passes over the same {n}-point batch, {b['seconds']:.1f} s at {best} threads; "
                  f"{len(counts)} thread counts tried for {slice_s:.1f} s each)",
        "threads_swept": {str(c): round(v["evals_per_s"], 1) for c, v in swept.items()},
        "single_thread_evals_per_s": swept[1]["evals_per_s"] if 1 in swept else None,
        "cgroup_cpu_share": host_cpu_share(),
        "host_logical_cpus": logical, "cpu_model": cpu_model(),
    }, float(err.max()), float(iters.mean())


class stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when a communicator is created; bench.py's stdout carries exactly one
    JSON line, so file descriptor 1 points at stderr while the communicator is being set up."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


class Watchdog:
    """A hung collective must not hang the scaling run.  Phases of the N > 1 run report progress with kick(); if
    nothing is reported for `limit` seconds the watchdog thread prints the best line this rank can still vouch for —
    on rank 0 the single-lane measurement if one completed, else a line that says `rccl-hung` — and leaves through
    os._exit: threads stuck inside RCCL cannot be joined, and a GPU process must never be re-exec'ed."""

    def __init__(self, rank, limit):
        import threading
        self.rank, self.limit = rank, limit
        self.last, self.phase = time.monotonic(), "start"
        self.fallback_line = None            # rank 0: JSON of a completed, verified measurement
        self.hung_line = None                # rank 0: JSON skeleton for the nothing-completed case
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def kick(self, phase=None):
        self.last = time.monotonic()
        if phase:
            self.phase = phase

    def _run(self):
        while not self._stop.wait(1.0):
            if time.monotonic() - self.last <= self.limit:
                continue
            print(f"[rank {self.rank}] watchdog: no progress for {self.limit:.0f} s in phase '{self.phase}'",
                  file=sys.stderr, flush=True)
            code = 3
            if self.rank == 0:
                if self.fallback_line is not None:
                    line = dict(self.fallback_line)
                    line["config"] = dict(line["config"], note=f"phase '{self.phase}' hung; the last completed measurement is reported")
                    print(json.dumps(line), flush=True)
                    code = 0
                elif self.hung_line is not None:
                    line = dict(self.hung_line)
                    line["config"] = dict(line["config"], allgather="rccl-hung", hung_phase=self.phase)
                    print(json.dumps(line), flush=True)
            sys.stdout.flush()
            os._exit(code)

    def stop(self):
        self._stop.set()


def device_count():
    import ctypes
    from evidence_amd import _abi
    n = ctypes.c_int32(0)
    _abi.load().rvll_device_count(ctypes.byref(n))
    return n.value


def kernel_source_sha():
    """sha256 over the kernel sources the committed PMC record was measured with (profiles/pmc_traffic.json)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("rvll_tile.h", "rvll_kernels.hip", "rvll_math.h"):
        h.update((REPO / "evidence_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()[:16]


def pmc_traffic(cfg, batch):
    """HBM bytes per launch of the log-L kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json, written by scripts/profile_gpu.sh + scripts/pmc_summary.py):
    FETCH_SIZE and WRITE_SIZE are in KiB; gfx950 reports half the bytes of a coalesced read stream, so
    the read side is doubled (MI355X_MICROARCH.md, HBM section).  The record is stamped with a hash of the kernel
    sources it was measured with: (None, reason) when it does not match the configuration or the sources."""
    path = REPO / "profiles" / "pmc_traffic.json"
    try:
        rec = json.loads(path.read_text())
    except (OSError, ValueError):
        return None, "no profiles/pmc_traffic.json"
    if rec.get("cfg") != cfg or rec.get("batch") != batch:
        return None, "profiles/pmc_traffic.json is for another configuration"
    if rec.get("kernel_source_sha") != kernel_source_sha():
        return None, "profiles/pmc_traffic.json was measured with other kernel sources (stale): traffic dropped"
    return ((2.0 * rec["fetch_kib"] + rec["write_kib"]) * 1024.0,
            f"profiles/pmc_traffic.json: {rec.get('source', '?')}")


def fip_extra(with_cpu):
    """SURVEY §8 f4 (never `value`): the FIP periodogram accumulation of fip_criterion.py:305-339 on synthetic
    posteriors (3 runs x 3 planet models x 30 000 samples, the script's 50 000 bins) — HIP-event kernel times,
    and the oracle's C fold of the same rows on one host core as the checker and CPU baseline."""
    from evidence_amd import fip
    rng = np.random.default_rng(2021)
    pmin, pmax, tobs, n = 1.5, 1000.0, 1000.0, 30000
    peaks = np.exp(rng.uniform(np.log(pmin * 2), np.log(pmax / 2), 3))
    post = [[None] + [(np.where(rng.random((n, k)) < 0.8, peaks[:k] * np.exp(rng.normal(0, 5e-4, (n, k))),
                                np.exp(rng.uniform(np.log(pmin), np.log(pmax), (n, k)))), rng.gamma(0.5, 1.0, n))
                      for k in (1, 2, 3)] for _ in range(3)]
    pky = rng.dirichlet(np.ones(4))
    _, nua, nub = fip.frequency_grid(pmin, pmax, tobs)
    got, t = fip.fip_periodogram(post, pky, nua, nub, repeats=10, return_timing=True)
    res = {"rows": int(t["rows"]), "bins": int(nua.size), "index_kernel_ms": t["index_ms"],
           "accumulate_kernel_ms": t["accumulate_ms"], "rows_per_s": t["rows"] / (t["index_ms"] + t["accumulate_ms"]) * 1e3}
    if with_cpu:
        from oracle import oracle          # checker + CPU baseline leg only
        periods, contrib, run_start = fip.flatten_posteriors(post, pky)
        t0 = time.perf_counter()
        want = oracle.fip_accumulate(nua, nub, periods, contrib, run_start)
        res["cpu_fold_ms_1core"] = (time.perf_counter() - t0) * 1e3
        res["bit_identical_to_oracle"] = bool(np.array_equal(got, want))
    return res


def run_extras(out, model, w, theta, B, with_cpu):
    """Extras of the N = 1 line (never `value`).  Each one is guarded: an extra that fails is reported under
    `extras_failed` and can never suppress the headline line."""
    def guarded(name, fn):
        try:
            fn()
        except Exception as exc:                                  # noqa: BLE001 — report, do not lose the line
            out.setdefault("extras_failed", {})[name] = f"{type(exc).__name__}: {exc}"
        finally:
            try:
                model.scalar_server(False)
                if model.dev_flip_lane() != 0:
                    model.dev_flip_lane()
                model.dev_upload_theta(theta)
            except Exception:                                     # noqa: BLE001
                pass

    def host_roundtrip():                 # PCIe-inclusive: host theta in, host log-L out
        model.log_likelihood_batch(theta)
        t1 = time.perf_counter()
        for _ in range(10):
            model.log_likelihood_batch(theta)
        out["host_roundtrip_evals_per_s"] = 10 * B / (time.perf_counter() - t1)

    def two_lanes():                      # two launches in flight on alternating pipeline lanes (how the N > 1 step
        for _ in range(100):              # overlaps its all-gather): independent batches hide each other's ramp and tail
            model.dev_loglike(B); model.dev_flip_lane()
        model.dev_sync()
        t1 = time.perf_counter()
        for _ in range(1000):
            model.dev_loglike(B); model.dev_flip_lane()
        model.dev_sync()
        out["two_lane_pipelined_evals_per_s"] = 1000 * B / (time.perf_counter() - t1)

    def prior_plus_loglike():             # cube -> theta -> log-L all on the device
        model.set_priors(w.priordict())
        def timed(n, step, reps, fill=True):
            if fill:
                model.dev_fill_cube(n, seed=99)
            for _ in range(5):
                step()
            model.dev_sync()
            t1 = time.perf_counter()
            for _ in range(reps):
                step()
            model.dev_sync()
            return reps * n / (time.perf_counter() - t1)
        out["prior_plus_loglike_evals_per_s"] = timed(B, lambda: (model.dev_prior(B), model.dev_loglike(B)), 50)
        # the same in ONE launch: the slim prior stage in front of the CU-wide log-L tile (rvll_dev_prior_loglike)
        out["prior_plus_loglike_one_launch_evals_per_s"] = timed(B, lambda: model.dev_prior_loglike(B), 50)
        # ... and the log-L kernel by itself on exactly those points (theta is resident from the call above): the
        # headline batch is another draw, and a launch's time depends on which points it gets
        out["loglike_alone_on_those_points_evals_per_s"] = timed(B, lambda: model.dev_loglike(B), 50, fill=False)
        # a sampler's proposal round: a small batch, where a launch is a large part of the step — the prior transform
        # in the log-L tile's staging step (one launch) against prior kernels + log-L kernel; a sync per step, as a
        # sampler that looks at every result would have
        small = 2048
        out["small_batch_prior_plus_loglike"] = {
            "points": small,
            "one_launch_evals_per_s": timed(small, lambda: (model.dev_prior_loglike(small), model.dev_sync()), 200),
            "two_launch_evals_per_s": timed(small, lambda: (model.dev_prior(small), model.dev_loglike(small), model.dev_sync()), 200)}

    def scalar_calls():                   # PolyChord's form, one theta per call: launch + sync vs the persistent kernel
        x0, lat = theta[0], {}
        for mode in ("launch", "server"):
            model.scalar_server(mode == "server")
            for _ in range(50):
                model.log_likelihood(x0)
            t1 = time.perf_counter()
            for _ in range(1000):
                model.log_likelihood(x0)
            lat[mode] = (time.perf_counter() - t1) / 1000 * 1e6
        out["scalar_call_us"] = {"launch_per_call": lat["launch"], "persistent_kernel": lat["server"]}
        # PolyChord's sequence per point: prior(cube), then loglike(of the theta it returned)
        from evidence_amd.callbacks import make_polychord_callbacks
        model.set_priors(w.priordict())
        cubes = w.sample_cube(256, seed=3)
        for name, kw in (("two_requests", {}), ("one_request", {"low_latency": True})):
            model.scalar_server(True)
            prior, loglike, _, _ = make_polychord_callbacks(model, **kw)
            for c in cubes[:32]:
                loglike(prior(c))
            t1 = time.perf_counter()
            for c in cubes:
                loglike(prior(c))
            out["scalar_call_us"]["polychord_pair_" + name] = (time.perf_counter() - t1) / len(cubes) * 1e6
        model.scalar_server(False)

    def nested_sampling():                # end to end with the proposal walk on the device
        from evidence_amd.callbacks import make_ultranest_callbacks, wrapped_params
        from evidence_amd.nested import run_nested_slice
        model.set_priors(w.priordict())
        vprior, vloglike = make_ultranest_callbacks(model, vectorized=True)
        inside = {"s": 0.0, "calls": 0, "slots": 0}

        def walker(*a):                   # the walk call by itself, and what it evaluated beyond the calls it reports
            t2 = time.perf_counter()
            res = model.slice_walk(*a)
            inside["s"] += time.perf_counter() - t2
            inside["calls"] += res[3]
            inside["slots"] += model.slice_walk_evaluated()
            return res

        t1 = time.perf_counter()
        ns = run_nested_slice(vprior, vloglike, model.ndim, nlive=32768, kbatch=16384, dlogz=1e-9,
                              max_calls=60_000_000, wrapped=wrapped_params(model.parnames), seed=1,
                              prior_loglike=model.prior_loglike_batch, walker=walker)
        out["nested_sampling_end_to_end"] = {"likelihood_calls_per_s": ns.ncall / (time.perf_counter() - t1),
                                             "calls": int(ns.ncall), "live_points": 32768, "deaths_per_iteration": 16384,
                                             "walk": "device (rvll_slice_walk)",
                                             "inside_walk_calls_per_s": inside["calls"] / inside["s"],
                                             "tile_slots_evaluated_per_call": inside["slots"] / max(1, inside["calls"])}

    def fip():
        out["fip_periodogram"] = fip_extra(with_cpu)

    for name, fn in (("host_roundtrip", host_roundtrip), ("two_lanes", two_lanes), ("prior_plus_loglike", prior_plus_loglike),
                     ("scalar_calls", scalar_calls), ("nested_sampling", nested_sampling), ("fip", fip)):
        guarded(name, fn)


def build_line(args, w, model, B, world, elapsed, gather, lanes, timed_region_kernel_ms):
    """The contract line (rank 0).  Returns (dict, gpu log-L of the resident batch, kernel seconds)."""
    value = world * B * args.steps / elapsed
    # dominant kernel, measured live with HIP events on the stream it is launched on
    tm = model.dev_time_loglike(B, warmup=max(3, args.warmup // 4), iters=max(10, min(args.steps, 200)))
    _, gpu_logl, _ = model.dev_download(B, flags=True)
    # the roofline uses the launch duration over the TIMED REGION itself (HIP events around its K launches on
    # their stream); the per-launch statistics of a separate event-per-launch run are reported beside it
    kern_s = (timed_region_kernel_ms if timed_region_kernel_ms is not None else tm["kernel_ms_mean"]) * 1e-3
    abytes = algorithmic_bytes_per_launch(w.ndim, w.table.n_epochs, B)
    achieved = abytes / kern_s / 1e9
    traffic, traffic_source = pmc_traffic(args.config, B)
    form = "CU-wide (one 1024-thread workgroup per CU)" if tm["threads"] == 1024 else "256-thread tiles"
    if world == 1:
        structure = "one stream: launch k+1 starts when launch k has drained (value == B / kernel time)"
    elif lanes > 1:
        structure = (f"{lanes} pipeline lanes (stream + communicator + buffers each): kernel ; all-gather in-stream, "
                     "consecutive steps alternate lanes, so two launches are in flight; the N=1 equivalent is "
                     "two_lane_pipelined_evals_per_s of the N=1 line")
    else:
        structure = "one lane: kernel ; all-gather in one stream, the next kernel starts behind the gather"
    out = {
        "metric": "live_point_logL_evals_per_sec", "value": value, "unit": "evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": {"fp64": "f64", "mixed": "f32 Newton / f64 phase+chi2 (NOT a parity mode)",
                                         "fp32": "f32 / f64 phase+sum (NOT a parity mode)"}[args.precision],
        "data": "synthetic",
        "config": {"workload": WORKLOAD_TEXT[args.config].format(b=B), "cfg": args.config,
                   "live_points_per_gpu": B, "epochs": w.table.n_epochs, "planets": len(model.layout.planets),
                   "instruments": len(w.table.insts), "free_parameters": w.ndim,
                   "parallelism": f"live-point shards x{world}", "allgather": gather, "lanes": lanes,
                   "kernel_form": form, "step_structure": structure, "runtime": GpuRVModel.runtime_info()},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "loglike_cu_kernel" if tm["threads"] == 1024 else "loglike_kernel",
                     "kernel_ms_timed_region": kern_s * 1e3,     # what `achieved` is computed from
                     "kernel_ms_mean": tm["kernel_ms_mean"],      # event-per-launch statistics (separate run)
                     "kernel_ms_min": tm["kernel_ms_min"], "kernel_ms_median": tm["kernel_ms_median"],
                     "algorithmic_bytes_per_launch": abytes,
                     "points_per_block": tm["points_per_block"], "blocks": tm["blocks"], "threads_per_block": tm["threads"],
                     "kernel_evals_per_s": B / kern_s,
                     "note": "fused kernel is fp64-VALU bound, not HBM bound (DESIGN.md); see valu_fp64"},
    }
    return out, gpu_logl, kern_s


def run_single(args, w, model, theta, B):
    # untimed pre-warm so that a short --warmup still starts from ramped clocks (the first few ms of
    # launches after idle run ~10 % slower); it precedes the W warm-up steps and is outside the timed region
    for _ in range(300):
        model.dev_loglike(B)
    model.dev_sync()
    for _ in range(args.warmup):
        model.dev_loglike(B)
    model.dev_sync()
    t0 = time.perf_counter()
    model.dev_mark(0)                            # HIP event on the stream the kernel is launched on
    for _ in range(args.steps):
        model.dev_loglike(B)
    model.dev_mark(1)
    model.dev_sync()
    elapsed = time.perf_counter() - t0
    out, gpu_logl, kern_s = build_line(args, w, model, B, 1, elapsed, "none", 1, model.dev_mark_elapsed_ms() / args.steps)
    if not args.no_extras:
        run_extras(out, model, w, theta, B, not args.no_cpu)
    if not args.no_cpu:
        cpu, perr, mean_it = cpu_baseline(w, model.layout, theta, gpu_logl, args.cpu_seconds)
        out["cpu_baseline"] = cpu
        out["parity_max_rel_err_vs_oracle"] = perr
        f_eval = flops_per_eval(len(model.layout.planets), w.table.n_epochs, mean_it)
        tf = B / kern_s * f_eval / 1e12
        out["roofline"]["valu_fp64"] = {"achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "frac": tf / FP64_VALU_PEAK_TFLOPS, "flops_per_eval": f_eval,
                                        "mean_newton_steps": mean_it,
                                        "kepler_solves_per_s": B / kern_s * len(model.layout.planets) * w.table.n_epochs,
                                        "newton_iterations_per_s": B / kern_s * len(model.layout.planets) * w.table.n_epochs * mean_it}
    print(json.dumps(out), flush=True)


def run_multi(args, w, model, theta, B, rank, world):
    """N > 1: one rank per GPU.  Order of events, each under the watchdog:
      0. ranks meet (evidence_amd/rendezvous.py); K steps are timed with the gather over the rendezvous sockets (a slower
         TRANSPORT, the same kernels) before RCCL is touched: the line of last resort;
      1. rank 0's RCCL id goes round, ONE communicator / lane per rank;
      2. a gathered step is checked (every rank finds its own log-L in its slot, everything finite);
      3. K steps are timed on one lane (barrier + sync on both sides, max over ranks): the next line to fall back on;
      4. the ranks try to add pipeline lanes (ncclCommSplit), agree on the minimum, check a gathered step per lane,
         and time K steps again — that is the line reported if it completes and verifies, else the one of step 3.
    If RCCL cannot be initialised on some rank (or hangs: the watchdog), the line of step 0 is the one reported, and
    config.allgather says `host-socket-fallback`."""
    from evidence_amd.rendezvous import Rendezvous
    wd = Watchdog(rank, float(os.environ.get("RVLL_WATCHDOG_S", "90")))
    with stdout_to_stderr():
        rdzv = Rendezvous.from_env(timeout=float(os.environ.get("RVLL_RDZV_TIMEOUT_S", "120")))
    wd.kick("socket transport")
    if rank == 0:
        wd.hung_line = {"metric": "live_point_logL_evals_per_sec", "value": 0.0, "unit": "evals/s", "n_gpus": world,
                        "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                        "config": {"workload": WORKLOAD_TEXT[args.config].format(b=B), "cfg": args.config,
                                   "parallelism": f"live-point shards x{world}"}}
    host_all = [None]
    gather = "host-socket-fallback"

    def step():
        model.dev_loglike(B)
        if gather == "rccl":
            model.allgather_logl(B)
        else:
            _, mine, _ = model.dev_download(B)
            host_all[0] = np.concatenate(rdzv.allgather(mine))

    def verified():
        """After a step: this rank's slot of the gathered vector is its own log-L, and every slot is finite."""
        model.dev_sync()
        _, mine, _ = model.dev_download(B)
        allv = model.download_gathered(world * B) if gather == "rccl" else host_all[0]
        good = int(np.array_equal(allv[rank * B:(rank + 1) * B], mine) and bool(np.isfinite(allv).all()))
        return rdzv.allreduce(good, "min") == 1

    def timed(label):
        wd.kick(f"{label}: warm-up")
        for i in range(args.warmup):
            step()
            if i % 64 == 0:
                wd.kick()
        model.dev_sync()
        rdzv.barrier()
        model.dev_sync()
        wd.kick(f"{label}: timed region")
        t0 = time.perf_counter()
        for i in range(args.steps):
            step()
            if i % 64 == 0:
                wd.kick()
        model.dev_sync()
        el = time.perf_counter() - t0
        wd.kick(f"{label}: reduce")
        rdzv.barrier()
        return rdzv.allreduce(el, "max")

    for _ in range(300):                             # untimed pre-warm (clocks), as in the N = 1 run
        model.dev_loglike(B)
    model.dev_sync()
    # 0. Before RCCL is touched at all: the same K steps with the gather over the rendezvous sockets.  Slower transport,
    #    same kernels - it exists so that a communicator that hangs in its set-up still leaves a measured line.
    step()
    if not verified():
        raise SystemExit(f"[rank {rank}] bench.py: the gathered log-L does not match the ranks' own values (sockets)")
    el_sock = timed("socket transport")
    if rank == 0:
        wd.fallback_line, _, _ = build_line(args, w, model, B, world, el_sock, gather, 1, None)

    wd.kick("communicator")
    ok, uid = 1, None
    with stdout_to_stderr():
        if rank == 0:
            try:
                uid = GpuRVModel.comm_unique_id()
            except Exception as exc:                  # noqa: BLE001 - reported, then the transport falls back
                print(f"[rank {rank}] RCCL unavailable: {exc}", file=sys.stderr)
        uid = rdzv.broadcast(uid, src=0)
        wd.kick()
        if uid is None:
            ok = 0
        else:
            try:
                model.comm_init(uid, world, rank)
            except Exception as exc:                  # noqa: BLE001
                ok = 0
                print(f"[rank {rank}] rvll_comm_init failed: {exc}", file=sys.stderr)
    gather = "rccl" if rdzv.allreduce(ok, "min") == 1 else "host-socket-fallback"
    if gather != "rccl" and ok:
        model.comm_destroy()
    elapsed, lanes, single = el_sock, 1, wd.fallback_line
    if gather == "rccl":
        wd.kick("first gathered step over RCCL")
        step()
        if not verified():
            raise SystemExit(f"[rank {rank}] bench.py: the gathered log-L does not match the ranks' own values")
        elapsed = timed("one lane")
        if rank == 0:
            single, _, _ = build_line(args, w, model, B, world, elapsed, gather, 1, None)
            wd.fallback_line = single
    want = int(os.environ.get("RVLL_LANES", "3"))
    if gather == "rccl" and want > 1:
        wd.kick("adding pipeline lanes")
        with stdout_to_stderr():
            have = model.comm_add_lanes(want)
        agreed = rdzv.allreduce(have, "min")          # every rank must cycle through the same communicators
        if agreed > 1:
            model.comm_set_lanes(agreed)
            wd.kick("first gathered steps on every lane")
            good = True
            for _ in range(agreed):                   # one gathered step per lane, each checked
                step()
                good = verified() and good
            if good:
                el2 = timed(f"{agreed} lanes")
                for _ in range(agreed):
                    step()
                if verified():
                    elapsed, lanes = el2, agreed
            if lanes == 1:
                model.comm_set_lanes(1)
    wd.kick("report")
    if rank == 0:
        out, _, _ = build_line(args, w, model, B, world, elapsed, gather, lanes, None)
        if lanes > 1:
            out["single_lane_evals_per_s"] = single["value"]
        print(json.dumps(out), flush=True)
    rdzv.barrier()
    wd.stop()
    if gather == "rccl":
        model.comm_destroy()
    rdzv.close()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    w = make_workload(args.config)
    B = args.batch or (CONFIGS[args.config]["batch"] // (8 if args.config in (4, 5) else 1))
    theta = w.sample_theta(B, seed=1234 + rank)
    ndev = device_count()
    if ndev < 1:
        sys.exit("bench.py: no HIP device visible; evidence_amd has no CPU path")
    if local_rank >= ndev:
        print(f"[rank {rank}] only {ndev} device(s) visible: sharing device {local_rank % ndev} "
              f"(rehearsal only - RCCL refuses two ranks on one GPU and the socket transport is used)", file=sys.stderr)
    model = GpuRVModel(w.fixedpardict, w.table, w.parnames, device=local_rank % ndev, precision=args.precision)
    if args.points_per_block:
        model.set_points_per_block(args.points_per_block)
    model.dev_upload_theta(theta)                # inputs resident in HBM before the timed region
    if world == 1:
        run_single(args, w, model, theta, B)
    else:
        run_multi(args, w, model, theta, B, rank, world)
    model.close()


if __name__ == "__main__":
    main()
