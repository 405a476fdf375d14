#!/usr/bin/env python3
"""Soak of the device-resident walk and the scalar-call server: for --seconds, random walks on changing configs
whose end points are re-evaluated from scratch (theta / log-L must match bit for bit, every end point above the
threshold), interleaved with scalar calls through the persistent kernel.  Run on the GPU box."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.callbacks import wrapped_params
from evidence_amd.synthetic import make_workload

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=60.0)
args = ap.parse_args()
rng = np.random.default_rng(123)
t_end = time.time() + args.seconds
walks = calls = scalars = bigs = lives = 0
forms = {}
models = {}
for cfg in (1, 2, 3, 5):
    w = make_workload(cfg)
    models[cfg] = (w, GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()))
last = time.time()
while time.time() < t_end:
    cfg = int(rng.choice([1, 2, 3, 3, 3, 5]))
    w, m = models[cfg]
    k = int(rng.integers(1, 6000 if cfg != 5 else 600))
    big = cfg != 5 and rng.random() < 0.12                   # more rows than walker slots: the row queue and the two-part
    if big:                                                  # launch come into play
        k = int(rng.integers(12000, 40000))
        bigs += 1
    m.set_walk_speculation(int(rng.choice([1, 2, 4, 4, 8])))
    # round 3: the form of the walk's second part — the queue (default), the rows form, its CU-wide and its 512-thread variants
    form = str(rng.choice(["0", "1", "2", "3"])) if big else "0"
    os.environ["RVLL_WALK_ROWS"] = form
    forms[form] = forms.get(form, 0) + (1 if big else 0)
    cube = rng.random((2 * k, m.ndim))
    theta, logl = m.prior_loglike_batch(cube)
    lstar = float(np.quantile(logl, rng.uniform(0.3, 0.6 if big else 0.98)))
    keep = logl > lstar
    if keep.sum() < 2:
        continue
    cube, theta, logl = cube[keep], theta[keep], logl[keep]
    d0 = cube - cube.mean(axis=0)
    chol = np.linalg.cholesky(d0.T @ d0 / max(1, len(cube) - 1) + 1e-10 * np.eye(m.ndim))
    c2, t2, l2, n = m.slice_walk(cube, theta, logl, lstar, chol, wrapped_params(m.parnames), nsteps=int(rng.integers(1, 30)),
                                 max_rounds=int(rng.choice([1, 3, 200])), seed=int(rng.integers(0, 2 ** 62)))
    th_chk, ll_chk = m.prior_loglike_batch(c2)
    assert (l2 > lstar).all(), "end point below the threshold"
    assert np.array_equal(th_chk, t2) and np.array_equal(ll_chk, l2), "end points do not describe the returned cubes"
    walks += 1; calls += n
    if big and rng.random() < 0.5:
        # the same walk through the resident live set (rvll_live_step): the walkers start from the first k of n rows, the factor is
        # handed over, the k rows with the lowest log-L are replaced — end points and count must be those of rvll_slice_walk
        nrow = len(cube)
        kd = min(nrow // 2, 16384)
        ll0 = m.live_init(cube)
        assert np.array_equal(ll0, logl)
        order = np.argsort(ll0, kind="stable")
        start = order[kd:][rng.integers(0, nrow - kd, kd)]
        ls = float(ll0[order[kd - 1]])
        seed2, nst = int(rng.integers(0, 2 ** 62)), int(rng.integers(1, 20))
        new, used = m.live_step(order, kd, start, ls, wrapped_params(m.parnames), nsteps=nst, seed=seed2, chol=chol)
        c3, t3, l3, n3 = m.slice_walk(cube[start], theta[start], logl[start], ls, chol, wrapped_params(m.parnames), nsteps=nst, seed=seed2)
        u_live, th_live, ll_live = m.live_get()
        assert used == n3 and np.array_equal(new, l3) and np.array_equal(u_live[order[:kd]], c3) and np.array_equal(th_live[order[:kd]], t3)
        assert np.array_equal(ll_live[order[kd:]], logl[order[kd:]])
        lives += 1
    m.scalar_server(True)
    for x, want in zip(t2[:20], l2[:20]):
        assert m.log_likelihood(x) == want, "scalar server disagrees"
        scalars += 1
    if rng.random() < 0.5:
        m.scalar_server(False)
    if time.time() - last > 20:
        last = time.time()
        print(f"... {walks} walks, {calls} likelihood calls, {scalars} scalar calls", flush=True)
for w, m in models.values():
    m.close()
print(f"soak ok: {walks} walks ({bigs} of them with more rows than walker slots: second part by form {dict(sorted(forms.items()))}; "
      f"{lives} repeated through the resident live set), {calls} likelihood calls inside walks, {scalars} scalar-server calls, all consistent")
