#!/usr/bin/env python3
"""Golden vectors for the FIP periodogram accumulation — runs ONLY in the build
container, where the reference checkout is mounted at /root/reference.

The reference's `evidence/fip_criterion.py` is a script that expects to sit in a directory of finished
nested-sampling runs (`<target>/<runid>/<target>_<x>_k<n>_<rep>/<same>.pkl` + `results.txt`) and writes
`fipnus/nu.txt` and `fipnus/fipnu_<target>_<runid>_maxpla<n>.txt` (its lines 296-340).  This generator
builds such a directory under build/ (git-ignored) from seeded synthetic posteriors — the pickles are plain
`types.SimpleNamespace` objects carrying the attributes the script reads — links the reference script into
it (a symlink; nothing is copied), runs it with the reference's interpreter environment, and stores the
inputs and the two output arrays as tests/golden/fip_<case>.npz.  Whatever the script does after it has
written the periodogram (plots) is irrelevant here and allowed to fail.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_fip_golden.py
"""
import datetime
import os
import pickle
import shutil
import subprocess
import sys
import types
from pathlib import Path

import numpy as np
import pandas as pd

HERE = Path(__file__).resolve().parent
REPO = HERE.parents[1]
REF_SCRIPT = Path("/root/reference/evidence/fip_criterion.py")
WORK = REPO / "build" / "fip_golden_tmp"


def make_case(name, seed, nmod, reps, pmin, pmax, tobs, nsamp, special=None):
    """Synthetic posteriors: posteriors[rep][k] = dict(samples [n,k], weights [n], logZ)."""
    rng = np.random.default_rng(seed)
    t0 = 55000.0
    times = np.sort(np.concatenate([[t0, t0 + tobs], rng.uniform(t0, t0 + tobs, 30)]))
    true_periods = np.exp(rng.uniform(np.log(pmin * 1.2), np.log(pmax * 0.8), nmod))
    post = []
    for rep in range(reps):
        per_k = []
        for k in range(nmod):
            logz = -500.0 + 12.0 * min(k, 2) - 1.5 * max(k - 2, 0) + rng.normal(0, 0.3)
            if k == 0:
                per_k.append({"logZ": logz})
                continue
            n = int(nsamp * (1 + 0.3 * rng.random()))
            s = np.empty((n, k))
            for j in range(k):
                # a concentrated mode plus a broad background, as nested-sampling posteriors have
                mode = true_periods[j] * np.exp(rng.normal(0, 2e-3, n))
                bg = np.exp(rng.uniform(np.log(pmin), np.log(pmax), n))
                s[:, j] = np.where(rng.random(n) < 0.7, mode, bg)
            w = rng.gamma(0.5, 1.0, n)                  # un-normalised, as read from the sampler output
            if special:
                special(rng, k, s, w, pmin, pmax)
            per_k.append({"logZ": logz, "samples": s, "weights": w})
        post.append(per_k)
    return dict(name=name, nmod=nmod, reps=reps, pmin=pmin, pmax=pmax, times=times, post=post)


def edge_special(rng, k, s, w, pmin, pmax):
    n = len(w)
    if k >= 2:
        s[: n // 4, 1] = s[: n // 4, 0] * (1 + rng.normal(0, 1e-4, n // 4))   # overlapping windows in one sample
        s[n // 4: n // 4 + 5, 1] = s[n // 4: n // 4 + 5, 0]                   # identical periods
    s[-1, 0] = pmin                      # exactly the last grid frequency
    s[-2, 0] = pmax                      # exactly the first grid frequency
    s[-3, 0] = pmin * 0.5                # beyond the grid on the high-frequency side
    s[-4, 0] = pmax * 3.0                # below the grid
    s[-5, 0] = pmax * 1.0000001
    s[-6, 0] = pmin * 0.9999999
    w[-7] = 0.0                          # a zero-weight sample
    w[: 3] *= 50.0                       # a few dominant samples


def write_tree(case, root):
    target, runid = "tgt" + case["name"], "run1"
    base = root / target / runid
    if base.exists():
        shutil.rmtree(base)
    base.mkdir(parents=True)
    datadict = {"instA": {"data": pd.DataFrame({"rjd": case["times"][::2]})},
                "instB": {"data": pd.DataFrame({"jdb": case["times"][1::2]})}}
    maxpla = case["nmod"] - 1
    rundict = {"prior_names": {f"planet{maxpla}_period": f"UniformFrequency: [{case['pmin']!r}, {case['pmax']!r}]"}}
    for rep, per_k in enumerate(case["post"]):
        for k, p in enumerate(per_k):
            run = f"{target}_x_k{k}_{rep}"
            (base / run).mkdir()
            names = sorted([f"planet{j + 1}_{q}" for j in range(k) for q in ("k1", "period", "ecc")] + ["instA_offset"])
            out = types.SimpleNamespace(sampler="PolyChord", base_dir="", logZ=p["logZ"],
                                        runtime=datetime.timedelta(seconds=60 + 7 * k + rep),
                                        parnames=names, datadict=datadict, rundict=rundict)
            n = 0
            if k:
                n = len(p["weights"])
                cols = np.zeros((n, len(names)))
                pidx = [i for i, nm in enumerate(names) if "period" in nm and "planet" in nm]
                cols[:, pidx] = p["samples"]
                out.posterior = types.SimpleNamespace(samples=cols, weights=p["weights"].copy())
            with open(base / run / (run + ".pkl"), "wb") as fh:
                pickle.dump(out, fh)
            (base / run / "results.txt").write_text(f"Nr. of samples in posterior: {max(n, 1)}\n")
    os.symlink(REF_SCRIPT, base / "fip_criterion.py")
    return base, target, runid, maxpla


def run_reference(base):
    env = dict(os.environ, PYTHONPATH="/root/reference", PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg")
    r = subprocess.run([sys.executable, "fip_criterion.py"], cwd=base, env=env, capture_output=True, text=True)
    return r


def main():
    cases = [
        make_case("small", 11, nmod=3, reps=2, pmin=1.5, pmax=200.0, tobs=320.0, nsamp=400),
        make_case("edges", 12, nmod=4, reps=3, pmin=1.1, pmax=500.0, tobs=61.0, nsamp=250, special=edge_special),
        make_case("single", 13, nmod=2, reps=1, pmin=2.0, pmax=50.0, tobs=1500.0, nsamp=600),
    ]
    WORK.mkdir(parents=True, exist_ok=True)
    for case in cases:
        base, target, runid, maxpla = write_tree(case, WORK)
        r = run_reference(base)
        fip_file = base / "fipnus" / f"fipnu_{target}_{runid}_maxpla{maxpla}.txt"
        if not fip_file.exists():
            sys.stderr.write(r.stdout[-2000:] + "\n" + r.stderr[-4000:])
            raise SystemExit(f"reference script produced no periodogram for case {case['name']}")
        fapnu = np.atleast_2d(np.loadtxt(fip_file))
        nu = np.loadtxt(base / "fipnus" / "nu.txt")
        arrays = {"nu": nu, "fapnu": fapnu, "times": case["times"],
                  "meta": np.array([case["nmod"], case["reps"]], dtype=np.int64),
                  "prange": np.array([case["pmin"], case["pmax"]])}
        for rep, per_k in enumerate(case["post"]):
            for k, p in enumerate(per_k):
                arrays[f"logZ_{rep}_{k}"] = np.float64(p["logZ"])
                if k:
                    arrays[f"samples_{rep}_{k}"] = p["samples"]
                    arrays[f"weights_{rep}_{k}"] = p["weights"]
        np.savez_compressed(HERE / f"fip_{case['name']}.npz", **arrays)
        touched = int((fapnu != 1.0).sum())
        print(f"{case['name']}: fapnu {fapnu.shape}, bins touched {touched}, min {fapnu.min():.3e}, "
              f"script exit {r.returncode}")
    shutil.rmtree(WORK)


if __name__ == "__main__":
    main()
