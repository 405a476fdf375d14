#!/usr/bin/env python3
"""bench.py — live-point log-L evaluations per second on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: one rank per GPU; live points shard across ranks, one RCCL all-gather of per-shard log-L per step.
     Launched either by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK /
     WORLD_SIZE / LOCAL_RANK in the environment) or just as written above: without WORLD_SIZE in the environment the
     process becomes a LAUNCHER that never touches the GPU — it starts N fresh children (subprocess, own process
     groups), gives them a private rendezvous address and secret, relays rank 0's one JSON line, and kills the
     groups and exits non-zero if a child fails or the run exceeds RVLL_LAUNCH_TIMEOUT_S.)

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
import importlib
import json
import os
import signal
import subprocess
import sys
import threading
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


NOMINAL_NEWTON_STEPS = 2.9     # mean Newton steps per solve at cfg3's priors (used until the oracle has counted them)


def roofline_block(nplanets, n_epochs, B, kern_s, mean_it, mean_it_source, traffic, traffic_source, hbm_gbs, abytes, tm):
    """`roofline` of the contract line: top level = the bound that governs the dominant kernel (fp64 VALU issue), the HBM view
    nested under "hbm"."""
    f_eval = flops_per_eval(nplanets, n_epochs, mean_it)
    tf = B / kern_s * f_eval / 1e12
    return {"bound": "fp64_valu", "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_VALU_PEAK_TFLOPS,
            "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC)", "traffic_source": traffic_source,
            "flops_per_eval": f_eval, "mean_newton_steps": mean_it, "mean_newton_steps_source": mean_it_source,
            "kepler_solves_per_s": B / kern_s * nplanets * n_epochs,
            "newton_iterations_per_s": B / kern_s * nplanets * n_epochs * mean_it,
            "kernel": "loglike_cu_kernel" if tm["threads"] == 1024 else "loglike_kernel",
            "kernel_ms_timed_region": kern_s * 1e3,     # what `achieved` is computed from
            "kernel_ms_mean": tm["kernel_ms_mean"],      # event-per-launch statistics (separate run)
            "kernel_ms_min": tm["kernel_ms_min"], "kernel_ms_median": tm["kernel_ms_median"],
            "points_per_block": tm["points_per_block"], "blocks": tm["blocks"], "threads_per_block": tm["threads"],
            "kernel_evals_per_s": B / kern_s,
            "hbm": {"achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_launch": abytes,
                    "note": "the fused kernel moves theta in + log-L out: 0.5 % of the HBM roofline, structurally (SURVEY 8d)"},
            "note": "no MFMA: elementwise + reduction; fp64 VALU issue governs (DESIGN.md section 4)"}


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
    on rank 0 the last completed, verified measurement, else a line that says `rccl-hung` — and leaves through
    os._exit: threads stuck inside RCCL cannot be joined, and a GPU process must never be re-exec'ed.

    RANK 0 REPORTS FIRST.  Every rank runs the same limit, so a peer's watchdog used to be able to fire a
    poll earlier, leave, and take rank 0 down with it (the launcher's SIGTERM, or a RendezvousError out of the next
    barrier) before rank 0 had printed the line it was holding.  Now the other ranks wait `grace` seconds longer than
    rank 0, and rank 0 leaves through report_and_exit() on EVERY way out: the watchdog, an exception from the
    rendezvous (run_multi catches it), SIGTERM from a launcher."""

    def __init__(self, rank, limit, grace=None):
        self.rank = rank
        grace = float(os.environ.get("RVLL_WATCHDOG_GRACE_S", "15")) if grace is None else grace
        self.limit = limit if rank == 0 else limit + grace
        self.last, self.phase = time.monotonic(), "start"
        self.fallback_line = None            # rank 0: JSON of a completed, verified measurement
        self.hung_line = None                # rank 0: JSON skeleton for the nothing-completed case
        self._stop = threading.Event()
        self._once = threading.Lock()
        self._asked = None                   # a signal handler's request to report and leave (request_exit)
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def kick(self, phase=None):
        self.last = time.monotonic()
        if phase:
            self.phase = phase

    def report_and_exit(self, why):
        """Print what this rank can vouch for and leave the process — at most once, from whichever thread gets here."""
        if not self._once.acquire(blocking=False):
            time.sleep(3600)                 # another thread is already on its way out through os._exit
        print(f"[rank {self.rank}] {why} (phase '{self.phase}')", file=sys.stderr, flush=True)
        code = 3
        if self.rank == 0:
            if self.fallback_line is not None:
                line = dict(self.fallback_line)
                line["config"] = dict(line["config"], note=f"{why} in phase '{self.phase}'; the last completed measurement is reported")
                print(json.dumps(line), flush=True)
                code = 0
            elif self.hung_line is not None:
                line = dict(self.hung_line)
                line["config"] = dict(line["config"], allgather="rccl-hung", hung_phase=self.phase, note=why)
                print(json.dumps(line), flush=True)
        sys.stdout.flush()
        os._exit(code)

    def request_exit(self, why):
        """For signal handlers: only leave a note — the watchdog thread reports and exits within a tenth of a second.  A
        handler runs on the main thread, which may itself be inside report_and_exit holding the lock (after a rendezvous
        error, say): calling report_and_exit from the handler then slept on the very thread that had the line to print, and
        the launcher's SIGKILL came before the line did."""
        self._asked = why

    def _run(self):
        while not self._stop.wait(0.1):
            if self._asked is not None:
                self.report_and_exit(self._asked)
            if time.monotonic() - self.last > self.limit:
                self.report_and_exit(f"watchdog: no progress for {self.limit:.0f} s")

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


def run_extras(out, model, w, theta, B, with_cpu, device_index=0):
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

    def large_host_batch():               # cube -> theta -> log-L through host arrays at 262144 rows (40 MB each way): the
        model.set_priors(w.priordict())   # streamed route (rvll_api.hip, stream_host_batch); a fresh cube array every call
        n, reps, tot = 262144, 10, 0.0
        src = w.sample_cube(n, seed=17)
        t_warm, warm = time.perf_counter(), 0
        while time.perf_counter() - t_warm < 0.3:      # (the CPU baseline before the extras left the GPU idle for seconds: clocks)
            model.prior_loglike_batch(src.copy())
            warm += 1
        for r in range(reps):
            cube = src.copy()
            t1 = time.perf_counter()
            model.prior_loglike_batch(cube)
            tot += time.perf_counter() - t1
        out["host_cube_to_logl_262144_rows"] = {"evals_per_s": reps * n / tot, "ms_per_call": tot / reps * 1e3, "calls_timed": reps,
                                                 "warm_up_calls": warm, "input": "a fresh 40 MB array every call"}
        model.dev_upload_theta(theta)     # (the resident batch of the extras that follow)

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
        # Round 4: a Kepler solve that wanders (e >= 0.97, tens to hundreds of Newton steps) is redone with correctly rounded
        # sin / cos by default (rvll_set_wander_exact; <= 1e-10 of the reference there too) — a serial chain of ~0.8 us steps
        # in one lane that the launch waits for.  Whether a batch holds such a point goes by the draw (this one: see
        # points_wandered; the headline batch holds none): the same three figures with the redo switched off.
        try:
            flags = model.dev_download(B, logl=False, flags=True)[2]
            model.set_wander_exact(False)
            out["prior_draw_with_exact_redo_off"] = {
                "points_wandered_in_this_draw": int(np.count_nonzero(flags & 4)),
                "loglike_alone_on_those_points_evals_per_s": timed(B, lambda: model.dev_loglike(B), 50, fill=False),
                "prior_plus_loglike_one_launch_evals_per_s": timed(B, lambda: model.dev_prior_loglike(B), 50),
                "prior_plus_loglike_evals_per_s": timed(B, lambda: (model.dev_prior(B), model.dev_loglike(B)), 50),
                "note": "rvll_set_wander_exact(0): flagged points (RVLL_FLAG_WANDERED) then agree with the reference to ~1e-9 instead of 1e-10"}
        finally:
            model.set_wander_exact(True)
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
        kw = dict(nlive=32768, kbatch=16384, dlogz=1e-9, max_calls=60_000_000, wrapped=wrapped_params(model.parnames), seed=1)

        def timed_run(**how):
            inside = {"s": 0.0, "calls": 0, "slots": 0}

            def walker(*a):               # the walk call by itself, and what it evaluated beyond the calls it reports
                t2 = time.perf_counter()
                res = model.slice_walk(*a)
                inside["s"] += time.perf_counter() - t2
                inside["calls"] += res[3]
                inside["slots"] += model.slice_walk_evaluated()
                return res

            class Live:                   # the same for the resident live set: time inside rvll_live_step
                live_init, live_get, live_dead, live_dead_count = model.live_init, model.live_get, model.live_dead, model.live_dead_count
                live_sort = model.live_sort

                @staticmethod
                def live_step(*a, **k):
                    t2 = time.perf_counter()
                    res = model.live_step(*a, **k)
                    inside["s"] += time.perf_counter() - t2
                    inside["calls"] += res[1]
                    inside["slots"] += model.slice_walk_evaluated()
                    return res

            t1 = time.perf_counter()
            if how.get("live"):
                ns = run_nested_slice(None, None, model.ndim, live=Live, **kw)
            else:
                ns = run_nested_slice(vprior, vloglike, model.ndim, prior_loglike=model.prior_loglike_batch, walker=walker, **kw)
            el = time.perf_counter() - t1
            return {"likelihood_calls_per_s": ns.ncall / el, "calls": int(ns.ncall), "seconds": el,
                    "inside_walk_calls_per_s": inside["calls"] / inside["s"],
                    "tile_slots_evaluated_per_call": inside["slots"] / max(1, inside["calls"]), "logz": ns.logz}

        res = timed_run(live=True)        # live set resident on the device: indices up, log-L down (rvll_live_step)
        res.update(live_points=32768, deaths_per_iteration=16384,
                   walk="device (rvll_live_step: live set, dead points and the walk resident in HBM)")
        res["host_managed_live_set"] = timed_run(live=False)      # round 2's form: rows through host buffers every iteration
        res["note"] = ("a THROUGHPUT loop: stopped at max_calls long before convergence (dlogz = 1e-9 never fires), ln Z is 'so far'; "
                       "converged runs are under nested_sampling_converged")
        res["walk_form"] = "rounds (step + batch log-L tiles per round, csrc/rvll_rounds.h)" if model.slice_walk_rounds() > 0 else "single kernel"
        out["nested_sampling_end_to_end"] = res

        # Converged evidence runs: dlogz = 0.5 — the reference's UltraNest default, evidence/ultranest/__init__.py:
        # 333-338 — with the live set resident on the device and host-managed, same seed.  cfg2 (one planet, 6 parameters): the
        # two forms agree within their errors.  cfg3 (three exchangeable planets, 19 parameters; the priors do not order the
        # periods, as SURVEY 8d has them): the posterior has 3! equivalent modes and aliases of each, and how many of them a
        # run's live points hold on to decides its ln Z — runs differ by tens of nats from seed to seed at 8192 live points
        # (profiles/r04_converged_runs.txt), which is a property of the sampler on this posterior, not of where it runs (the
        # reference breaks the symmetry with PolyChord's sorted priors: evidence/priors.py:462-467).
        def converged(wl, mdl, nlive, kbatch, seed):
            pr, ll = make_ultranest_callbacks(mdl, vectorized=True)
            got = {}
            for name, how in (("resident", dict(live=mdl)), ("host_managed", dict(walker=mdl.slice_walk, prior_loglike=mdl.prior_loglike_batch))):
                t1 = time.perf_counter()
                ns = run_nested_slice(pr, ll, mdl.ndim, nlive=nlive, kbatch=kbatch, dlogz=0.5, max_calls=2_000_000_000,
                                      wrapped=wrapped_params(mdl.parnames), seed=seed, **how)
                el = time.perf_counter() - t1
                got[name] = {"logz": ns.logz, "logzerr": ns.logzerr, "information": ns.information, "iterations": int(ns.niter),
                             "calls": int(ns.ncall), "seconds": el, "calls_per_s": ns.ncall / el}
            d, s = abs(got["resident"]["logz"] - got["host_managed"]["logz"]), float(np.hypot(got["resident"]["logzerr"], got["host_managed"]["logzerr"]))
            got.update(live_points=nlive, deaths_per_iteration=kbatch, dlogz=0.5, seed=seed, abs_difference=d, combined_sigma=s,
                       agree_within_3_sigma=bool(d <= 3 * s))
            return got

        conv = {"cfg3": converged(w, model, 8192, 2048, 1)}
        try:
            w2 = make_workload(2)
            cls, _ = model_class()
            with cls(w2.fixedpardict, w2.table, w2.parnames, priordict=w2.priordict(), device=device_index) as m2:
                conv["cfg2"] = converged(w2, m2, 8192, 2048, 1)
        except Exception as exc:                                  # noqa: BLE001 — an extra never costs the line
            conv["cfg2_failed"] = f"{type(exc).__name__}: {exc}"
        conv["note"] = ("cfg3's exchangeable planets make its evidence a matter of how many of the 3! modes (and their aliases) a run holds: "
                        "seed-to-seed scatter of tens of nats at this size; see profiles/r04_converged_runs.txt")
        out["nested_sampling_converged"] = conv

    def fip():
        out["fip_periodogram"] = fip_extra(with_cpu)

    for name, fn in (("host_roundtrip", host_roundtrip), ("large_host_batch", large_host_batch), ("two_lanes", two_lanes), ("prior_plus_loglike", prior_plus_loglike),
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
    elif gather != "rccl":
        structure = ("socket transport: kernel ; synchronous download of the shard's log-L ; all-gather over the rendezvous "
                     "sockets (star through rank 0) - fully synchronous per step, no overlap; the fallback when RCCL "
                     "cannot be initialised (e.g. ranks sharing one device)")
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
                   "kernel_form": form, "step_structure": structure, "runtime": type(model).runtime_info(),
                   "torch_in_process": "torch" in sys.modules,
                   "launched_by": "bench.py itself" if os.environ.get("RVLL_SELF_LAUNCHED") else
                                  ("an external launcher (RANK / WORLD_SIZE in the environment)" if world > 1 else "directly")},
        # The bound that governs this kernel is fp64 VALU issue (software sin / cos, division, log: no MFMA, and 0.5 % of the HBM
        # roofline — SURVEY 8d), so THAT is what the top level of `roofline` reports: algorithmic FLOP by the
        # SURVEY's convention F_eval = Ne (Np (128 + 96 n_it) + 60) per launch / the launch duration, against the 78.6 TFLOP/s
        # vector fp64 peak.  n_it: the Newton steps per solve the oracle counts on this batch when the CPU leg runs
        # (run_single fills it in), the SURVEY's nominal 2.9 until then.  `traffic` stays HBM bytes per launch from the PMC
        # record; the HBM view of the same launch (algorithmic bytes, GB/s, fraction of 8 TB/s) is nested under `hbm`.
        "roofline": roofline_block(len(model.layout.planets), w.table.n_epochs, B, kern_s, NOMINAL_NEWTON_STEPS, "nominal (SURVEY 8d)",
                                   traffic, traffic_source, achieved, abytes, tm),
    }
    _, stub = model_class()
    if stub:                                     # the CPU tests of the launch path: never to be read as a measurement
        out["config"]["stub_model"] = stub
        out["data"] = "stub model (test harness of the launch path; NOT a measurement)"
    return out, gpu_logl, kern_s


def model_class():
    """GpuRVModel — or, for the CPU tests of the launch path only, a stand-in named by RVLL_BENCH_MODEL_FOR_TESTS
    ("module:Class").  A line produced with a stand-in says so (`config.stub_model`, `data`) and is not a measurement."""
    spec = os.environ.get("RVLL_BENCH_MODEL_FOR_TESTS")
    if not spec:
        return GpuRVModel, None
    mod, _, cls = spec.partition(":")
    return getattr(importlib.import_module(mod), cls), spec


def shader_clocks_mhz():
    """Current shader clock of every amdgpu card that exposes one (sysfs pp_dpm_sclk, the level marked '*'), keyed by
    PCI address.  Read while launches are in flight, outside the timed region; {} where sysfs says nothing."""
    out = {}
    try:
        import glob
        for path in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
            try:
                dev = os.path.basename(os.path.realpath(os.path.dirname(path)))
                for line in open(path):
                    if line.rstrip().endswith("*"):
                        out[dev] = int("".join(ch for ch in line.split(":", 1)[1] if ch.isdigit()))
            except (OSError, ValueError, IndexError):
                continue
    except Exception:                                             # noqa: BLE001 — a diagnostic must never cost the line
        pass
    return out


def prewarm(model, B, min_seconds=0.3, max_seconds=3.0, block=50, agree=0.01, kick=None):
    """Untimed pre-warm BY TIME AND CONVERGENCE (a fixed 300 launches = 20 ms left the driver's
    20-step run on clocks that were still coming up, 7.5 % below the steady state the 2000-step runs measure).
    Launches go out in blocks of `block`, each block timed on the wall clock with a sync behind it, until at least
    `min_seconds` have passed AND the last two blocks agree within `agree` (or `max_seconds`, which is then
    reported as not converged).  It precedes the W warm-up steps and is outside the timed region.  The shader clock
    is read from sysfs while a further block is in flight.  (RVLL_PREWARM_S="min,max" overrides the two durations.)"""
    if os.environ.get("RVLL_PREWARM_S"):
        min_seconds, max_seconds = (float(x) for x in os.environ["RVLL_PREWARM_S"].split(","))
    t0 = time.perf_counter()
    blocks, launches, converged = [], 0, False
    while True:
        t1 = time.perf_counter()
        for _ in range(block):
            model.dev_loglike(B)
        model.dev_sync()
        now = time.perf_counter()
        blocks.append((now - t1) / block)
        launches += block
        if kick:
            kick()
        if len(blocks) >= 2 and abs(blocks[-1] - blocks[-2]) <= agree * blocks[-1]:
            converged = True
            if now - t0 >= min_seconds:
                break
        else:
            converged = False
        if now - t0 >= max_seconds:
            break
    for _ in range(block):
        model.dev_loglike(B)
    clocks = shader_clocks_mhz()
    model.dev_sync()
    return {"seconds": time.perf_counter() - t0, "launches": launches + block, "converged": converged,
            "first_block_us_per_launch": blocks[0] * 1e6, "last_block_us_per_launch": blocks[-1] * 1e6,
            "rule": f">= {min_seconds} s and two consecutive {block}-launch blocks within {agree:.0%} (cap {max_seconds} s)",
            "shader_clock_mhz_under_load": clocks or None}


def sharded_config_steps(make, rank, ranks, shards, steps, step_of, kick=None, verify=None):
    """BASELINE.json configs[3] and configs[4] as steps (extras, never `value`): the 65 536-point 3-planet 1000-epoch
    batch and the 131 072-point 5-planet + drift 2000-epoch batch, each cut into `shards` shards (8: 8192 and 16 384
    live points per GPU) of which `ranks` run here, one per GPU, the latter in fp64 and in the two reduced-precision
    modes with the error between them on this rank's shard.  `make(cfg, precision) -> (workload, model)`;
    `step_of(model, B) -> (step, finish)`: the callable of one step (kernel, plus the all-gather when there is more
    than one rank) and the reduction of the elapsed time over ranks; `verify(model, B)` checks a gathered step."""
    out = {}
    for cfg, total in ((4, CONFIGS[4]["batch"]), (5, CONFIGS[5]["batch"])):
        B = total // shards
        ref_logl, entry = None, {"live_points_total": B * ranks, "live_points_per_gpu": B, "shards_of_the_config": shards}
        for prec in (("fp64",) if cfg == 4 else ("fp64", "mixed", "fp32")):
            w, model = make(cfg, prec)
            try:
                theta = w.sample_theta(B, seed=4321 + rank)
                model.dev_upload_theta(theta)
                step, finish = step_of(model, B)
                n = max(20, min(steps, 50 if cfg == 4 else 20))
                # a fresh handle: the clocks fell while it was being set up.  One rank warms up by time; several must all
                # make the same number of steps (every step holds a collective)
                t_warm, warm_steps = time.perf_counter(), 0
                while (time.perf_counter() - t_warm < 0.1) if ranks == 1 else (warm_steps < 60):
                    for _ in range(3):
                        step()
                    warm_steps += 3
                    model.dev_sync()
                    if kick:
                        kick()
                if verify and not verify(model, B):
                    raise RuntimeError(f"cfg{cfg} {prec}: the gathered log-L does not match the ranks' own values")
                if kick:
                    kick()
                t0 = time.perf_counter()
                for _ in range(n):
                    step()
                model.dev_sync()
                el = finish(time.perf_counter() - t0)
                _, logl, _ = model.dev_download(B)
                rec = {"ms_per_step": el / n * 1e3, "evals_per_s": ranks * B * n / el, "steps": n,
                       "kepler_solves_per_s": ranks * B * n / el * len(model.layout.planets) * w.table.n_epochs}
                if prec == "fp64":
                    ref_logl = logl
                    # the same steps without the exact redo of wandering solves (round 4, rvll_set_wander_exact: on by default):
                    # what a launch waits for when its shard holds such a point (every rank makes the same steps: collectives)
                    flags = model.dev_download(B, logl=False, flags=True)[2]
                    model.set_wander_exact(False)
                    for _ in range(3):
                        step()
                    model.dev_sync()
                    if kick:
                        kick()
                    t0 = time.perf_counter()
                    for _ in range(n):
                        step()
                    model.dev_sync()
                    el_off = finish(time.perf_counter() - t0)
                    model.set_wander_exact(True)
                    rec["exact_redo_off"] = {"ms_per_step": el_off / n * 1e3, "points_wandered_in_this_shard": int(np.count_nonzero(flags & 4))}
                else:
                    err = np.abs(logl - ref_logl) / np.maximum(np.abs(ref_logl), 1e-300)
                    rec.update(max_rel_err_vs_fp64=float(err.max()), median_rel_err_vs_fp64=float(np.median(err)),
                               max_abs_err_vs_fp64=float(np.abs(logl - ref_logl).max()))
                if cfg == 4:
                    entry.update(rec)
                else:
                    entry[prec] = rec
            finally:
                model.close()
            if kick:
                kick()
        entry["workload"] = WORKLOAD_TEXT[cfg].format(b=B)
        out[f"cfg{cfg}"] = entry
    return out


def run_single(args, w, model, theta, B):
    warm = prewarm(model, B)
    for _ in range(args.warmup):
        model.dev_loglike(B)
    model.dev_sync()
    model.dev_mark(0)                            # HIP event on the stream the kernel is launched on (not one of the K steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.dev_loglike(B)
    model.dev_mark(1)
    model.dev_sync()
    elapsed = time.perf_counter() - t0
    out, gpu_logl, kern_s = build_line(args, w, model, B, 1, elapsed, "none", 1, model.dev_mark_elapsed_ms() / args.steps)
    out["config"]["prewarm"] = warm
    if not args.no_extras:
        run_extras(out, model, w, theta, B, not args.no_cpu, args.device_index)
        cls, _ = model_class()
        try:
            def make(cfg, prec):
                ww = make_workload(cfg)
                return ww, cls(ww.fixedpardict, ww.table, ww.parnames, device=args.device_index, precision=prec)
            # one GPU, one of the eight shards BASELINE.json cuts these configurations into (rates are this GPU's)
            out["sharded_configs_one_of_8_shards"] = sharded_config_steps(
                make, 0, 1, 8, args.steps, lambda m, b: ((lambda: m.dev_loglike(b)), (lambda el: el)))
            model.dev_upload_theta(theta)
        except Exception as exc:                                  # noqa: BLE001 — an extra never costs the line
            out.setdefault("extras_failed", {})["sharded_configs"] = f"{type(exc).__name__}: {exc}"
    if not args.no_cpu:
        cpu, perr, mean_it = cpu_baseline(w, model.layout, theta, gpu_logl, args.cpu_seconds)
        out["cpu_baseline"] = cpu
        out["parity_max_rel_err_vs_oracle"] = perr
        r = out["roofline"]                          # the Newton steps the oracle counted on this very batch replace the nominal figure
        out["roofline"] = roofline_block(len(model.layout.planets), w.table.n_epochs, B, kern_s, mean_it, "counted by the oracle on this batch",
                                         r["traffic"], r["traffic_source"], r["hbm"]["achieved"], r["hbm"]["algorithmic_bytes_per_launch"],
                                         {"threads": r["threads_per_block"], "kernel_ms_mean": r["kernel_ms_mean"], "kernel_ms_min": r["kernel_ms_min"],
                                          "kernel_ms_median": r["kernel_ms_median"], "points_per_block": r["points_per_block"], "blocks": r["blocks"]})
    print(json.dumps(out), flush=True)


class CommSetup:
    """RCCL communicator of one model over the rendezvous: rank 0's id goes round, every rank initialises, the ranks
    agree (MIN) whether all of them could.  Returns "rccl" or "host-socket-fallback"."""

    @staticmethod
    def establish(cls, model, rdzv, rank, world, wd):
        ok, uid = 1, None
        with stdout_to_stderr():
            if rank == 0:
                try:
                    uid = cls.comm_unique_id()
                except Exception as exc:              # noqa: BLE001 - reported, then the transport falls back
                    print(f"[rank {rank}] RCCL unavailable: {exc}", file=sys.stderr)
            uid = rdzv.broadcast(uid, src=0)
            wd.kick()
            if uid is None:
                ok = 0
            else:
                try:
                    model.comm_init(uid, world, rank)
                except Exception as exc:              # noqa: BLE001
                    ok = 0
                    print(f"[rank {rank}] rvll_comm_init failed: {exc}", file=sys.stderr)
        agreed = rdzv.allreduce(ok, "min") == 1
        if not agreed and ok:
            model.comm_destroy()
        return "rccl" if agreed else "host-socket-fallback"


def run_multi(args, w, model, theta, B, rank, world, device_index):
    """N > 1: one rank per GPU.  Order of events, each under the watchdog:
      0. ranks meet (evidence_amd/rendezvous.py); K steps are timed with the gather over the rendezvous sockets (a slower
         TRANSPORT, the same kernels) before RCCL is touched: the line of last resort;
      1. rank 0's RCCL id goes round, ONE communicator / lane per rank;
      2. a gathered step is checked (every rank finds its own log-L in its slot, everything finite);
      3. K steps are timed on one lane (barrier + sync on both sides, max over ranks): the next line to fall back on;
      4. the ranks try to add pipeline lanes (ncclCommSplit), agree on the minimum, check a gathered step per lane,
         and time K steps again — that is the line reported if it completes and verifies, else the one of step 3;
      5. extras (never `value`): BASELINE.json configs[3] / configs[4] as sharded steps over the same transport.
    If RCCL cannot be initialised on some rank (or hangs: the watchdog), the line of step 0 is the one reported, and
    config.allgather says `host-socket-fallback`.  Rank 0 leaves through the watchdog's report_and_exit on every
    failure path — a peer that died, a rendezvous error, SIGTERM from a launcher — so a line it holds is printed."""
    from evidence_amd.rendezvous import Rendezvous, RendezvousError
    cls, stub = model_class()
    wd = Watchdog(rank, float(os.environ.get("RVLL_WATCHDOG_S", "90")))
    if rank == 0:
        wd.hung_line = {"metric": "live_point_logL_evals_per_sec", "value": 0.0, "unit": "evals/s", "n_gpus": world,
                        "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                        "config": {"workload": WORKLOAD_TEXT[args.config].format(b=B), "cfg": args.config,
                                   "parallelism": f"live-point shards x{world}"}}
        signal.signal(signal.SIGTERM, lambda *_: wd.request_exit("SIGTERM (a peer or the launcher ended the run)"))
    try:
        _run_multi_body(args, w, model, theta, B, rank, world, device_index, cls, stub, wd, Rendezvous)
    except (RendezvousError, OSError) as exc:
        # a peer is gone (its watchdog, a crash): whatever this rank completed and verified is still true
        wd.report_and_exit(f"{type(exc).__name__}: {exc}")


def _run_multi_body(args, w, model, theta, B, rank, world, device_index, cls, stub, wd, Rendezvous):
    with stdout_to_stderr():
        rdzv = Rendezvous.from_env(timeout=float(os.environ.get("RVLL_RDZV_TIMEOUT_S", "120")))
    wd.kick("socket transport")
    devices = rdzv.allgather(int(device_index))
    shared = len(set(devices)) < world           # single node: two ranks on one device index share a GPU (rehearsal)
    host_all = [None]
    state = {"gather": "host-socket-fallback"}

    def step_of(m, b):
        """One step of model m over b resident points with the CURRENT transport of the main model's run; for the
        extras' models the RCCL transport needs their own communicator (CommSetup)."""
        def step():
            m.dev_loglike(b)
            if state["gather"] == "rccl":
                m.allgather_logl(b)
            else:
                _, mine, _ = m.dev_download(b)
                host_all[0] = np.concatenate(rdzv.allgather(mine))
        return step

    step = step_of(model, B)

    def verified(m=model, b=B):
        """After a step: this rank's slot of the gathered vector is its own log-L, and every slot is finite."""
        m.dev_sync()
        _, mine, _ = m.dev_download(b)
        allv = m.download_gathered(world * b) if state["gather"] == "rccl" else host_all[0]
        good = int(np.array_equal(allv[rank * b:(rank + 1) * b], mine) and bool(np.isfinite(allv).all()))
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

    def line(elapsed, lanes):
        out, _, _ = build_line(args, w, model, B, world, elapsed, state["gather"], lanes, None)
        out["config"].update(prewarm=warm, devices=devices, shared_device=shared)
        if shared:
            out["data"] += " (REHEARSAL: ranks share a device, this is not an N-GPU measurement)"
        return out

    warm = prewarm(model, B, kick=wd.kick)            # untimed pre-warm (clocks), as in the N = 1 run
    # 0. Before RCCL is touched at all: the same K steps with the gather over the rendezvous sockets.  Slower transport,
    #    same kernels - it exists so that a communicator that hangs in its set-up still leaves a measured line.
    step()
    if not verified():
        raise SystemExit(f"[rank {rank}] bench.py: the gathered log-L does not match the ranks' own values (sockets)")
    el_sock = timed("socket transport")
    if rank == 0:
        wd.fallback_line = line(el_sock, 1)

    wd.kick("communicator")
    state["gather"] = CommSetup.establish(cls, model, rdzv, rank, world, wd)
    elapsed, lanes, single = el_sock, 1, wd.fallback_line
    if state["gather"] == "rccl":
        wd.kick("first gathered step over RCCL")
        step()
        if not verified():
            raise SystemExit(f"[rank {rank}] bench.py: the gathered log-L does not match the ranks' own values")
        elapsed = timed("one lane")
        if rank == 0:
            single = line(elapsed, 1)
            wd.fallback_line = single
    want = int(os.environ.get("RVLL_LANES", "3"))
    if state["gather"] == "rccl" and want > 1:
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
    out = None
    if rank == 0:
        out = line(elapsed, lanes)
        if lanes > 1:
            out["single_lane_evals_per_s"] = single["value"]
        wd.fallback_line = out                        # from here on only extras can go wrong
    # 5. the two BASELINE configurations that name 8 GPUs, as sharded steps (extras of this line)
    if not args.no_extras:
        wd.kick("sharded configs (extras)")
        main_gather = state["gather"]
        try:
            models = []

            def make(cfg, prec):
                ww = make_workload(cfg)
                m = cls(ww.fixedpardict, ww.table, ww.parnames, device=device_index, precision=prec)
                models.append(m)
                if main_gather == "rccl":             # a communicator of its own; all ranks fall back together if one cannot
                    state["gather"] = CommSetup.establish(cls, m, rdzv, rank, world, wd)
                return ww, m

            def steps_of(m, b):
                def finish(el):
                    rdzv.barrier()
                    return rdzv.allreduce(el, "max")
                return step_of(m, b), finish

            extra = sharded_config_steps(make, rank, world, world, args.steps, steps_of, kick=wd.kick, verify=verified)
            extra["allgather"] = state["gather"]
            if rank == 0:
                out["sharded_configs"] = extra
        except (SystemExit, KeyboardInterrupt):
            raise
        except Exception as exc:                      # noqa: BLE001 — an extra never costs the line ...
            from evidence_amd.rendezvous import RendezvousError
            if isinstance(exc, (RendezvousError, OSError)):
                raise                                 # ... but a lost peer ends the run through report_and_exit
            if rank == 0:
                out.setdefault("extras_failed", {})["sharded_configs"] = f"{type(exc).__name__}: {exc}"
        state["gather"] = main_gather
    wd.kick("report")
    if rank == 0:
        print(json.dumps(out), flush=True)
    rdzv.barrier()
    wd.stop()
    if state["gather"] == "rccl":
        model.comm_destroy()
    rdzv.close()


def launch_self(args):
    """`python bench.py --gpus N` with no launcher around it (VERDICT r2 missing #1): this process becomes the launcher.
    It never makes a HIP call (a GPU process must not fork / exec others), starts N fresh interpreters on this very
    script — RANK / LOCAL_RANK / WORLD_SIZE, a private rendezvous address and a random secret for the frame MACs in
    their environment, each in its own process group —, relays rank 0's JSON line, and ends the run when it should end:
      * rank 0 exits            -> the others get a few seconds to follow, then their groups are killed;
      * another rank fails      -> rank 0 gets its watchdog's grace to print what it holds (it notices the lost peer by
                                   itself through the rendezvous), then everything is killed;
      * RVLL_LAUNCH_TIMEOUT_S   -> everything is killed.
    Exit code 0 iff rank 0 printed a line and exited 0 and no rank failed by itself.  This is the MPI launcher the reference leaves to its samplers
    (evidence/polychord/__init__.py:21-29,176-199), reduced to what one node needs."""
    n = args.gpus
    secret = os.urandom(32).hex()
    address = f"unix:rvll-bench-{os.getpid()}-{secret[:16]}"
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), RVLL_RDZV=address, RVLL_RDZV_SECRET=secret,
                RVLL_SELF_LAUNCHED="1", MASTER_ADDR="127.0.0.1")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    script = str(Path(__file__).resolve())
    children = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        children.append(subprocess.Popen([sys.executable, script] + sys.argv[1:], env=env, text=True,
                                         stdout=subprocess.PIPE if r == 0 else sys.stderr, start_new_session=True))
    lines = []

    def relay():                                       # rank 0's stdout: JSON lines are kept, anything else goes to stderr
        for raw in children[0].stdout:
            text = raw.strip()
            try:
                if text.startswith("{") and "metric" in json.loads(text):
                    lines.append(text)
                    continue
            except ValueError:
                pass
            if text:
                print(text, file=sys.stderr, flush=True)

    reader = threading.Thread(target=relay, daemon=True)
    reader.start()

    def kill_all():
        for sig, wait in ((signal.SIGTERM, 5.0), (signal.SIGKILL, 5.0)):
            alive = [p for p in children if p.poll() is None]
            if not alive:
                return
            for p in alive:
                try:
                    os.killpg(p.pid, sig)              # exactly the groups started above
                except (ProcessLookupError, PermissionError):
                    pass
            t_end = time.monotonic() + wait
            while time.monotonic() < t_end and any(p.poll() is None for p in alive):
                time.sleep(0.05)

    stop = {"sig": None}
    for sig in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sig, lambda s, *_: stop.__setitem__("sig", s))
    deadline = time.monotonic() + float(os.environ.get("RVLL_LAUNCH_TIMEOUT_S", "1500"))
    grace = float(os.environ.get("RVLL_WATCHDOG_GRACE_S", "15")) + 5.0
    why, failed_at = None, None
    while True:
        codes = [p.poll() for p in children]
        now = time.monotonic()
        if codes[0] is not None:                       # rank 0 is done (either way): the others follow or are ended
            t_end = now + 10.0
            while time.monotonic() < t_end and any(p.poll() is None for p in children):
                time.sleep(0.05)
            break
        if failed_at is None and any(c not in (None, 0) for c in codes[1:]):
            failed_at = now
            why = next(f"rank {i} exited with code {c}" for i, c in enumerate(codes) if c not in (None, 0))
        if failed_at is not None and now - failed_at > grace:
            break
        if stop["sig"] is not None:
            why = f"signal {stop['sig']}"
            break
        if now > deadline:
            why = "RVLL_LAUNCH_TIMEOUT_S exceeded"
            break
        time.sleep(0.1)
    own_exit = [p.poll() for p in children]            # before the launcher ends anybody: how each rank left by itself
    kill_all()
    reader.join(timeout=5.0)
    rc0 = children[0].poll()
    bad = [(i, c) for i, c in enumerate(own_exit) if c not in (None, 0)]
    if bad and (why is None or why.startswith("rank ")):
        why = "; ".join(f"rank {i} exited with code {c}" for i, c in bad)
    if why:
        print(f"bench.py launcher: {why}; ranks ended with {[p.poll() for p in children]}", file=sys.stderr, flush=True)
    if lines:
        print(lines[-1], flush=True)
    if lines and rc0 == 0 and why is None:
        return 0
    # a line may have gone out above (rank 0 reports what it completed and verified before a peer was lost), but a run
    # in which a rank failed, or which had to be ended from here, does not exit 0
    return rc0 if rc0 not in (None, 0) else 4


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_self(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world                        # a launcher's WORLD_SIZE is the truth

    cls, stub = model_class()
    w = make_workload(args.config)
    B = args.batch or (CONFIGS[args.config]["batch"] // (8 if args.config in (4, 5) else 1))
    theta = w.sample_theta(B, seed=1234 + rank)
    ndev = cls.device_count() if stub else device_count()
    if ndev < 1:
        sys.exit("bench.py: no HIP device visible; evidence_amd has no CPU path")
    if local_rank >= ndev:
        print(f"[rank {rank}] only {ndev} device(s) visible: sharing device {local_rank % ndev} "
              f"(rehearsal only - RCCL refuses two ranks on one GPU and the socket transport is used)", file=sys.stderr)
    args.device_index = local_rank % ndev
    model = cls(w.fixedpardict, w.table, w.parnames, device=args.device_index, precision=args.precision)
    if args.points_per_block:
        model.set_points_per_block(args.points_per_block)
    model.dev_upload_theta(theta)                # inputs resident in HBM before the timed region
    if world == 1:
        run_single(args, w, model, theta, B)
    else:
        run_multi(args, w, model, theta, B, rank, world, args.device_index)
    model.close()


if __name__ == "__main__":
    main()
