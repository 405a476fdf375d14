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
walks = calls = scalars = bigs = lives = crosses = 0
forms = {}
hows = {}
ROUNDS_KNOBS = ("RVLL_WALK_ROUNDS", "RVLL_ROUNDS_GROUPS", "RVLL_ROUNDS_FREE", "RVLL_ROUNDS_DEPTH", "RVLL_ROUNDS_FORM", "RVLL_ROUNDS_PB", "RVLL_ROUNDS_W")
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
    # round 4: which form walks — by size (the default), the rounds form forced with its knobs drawn at random, or the single-kernel form
    for key in ROUNDS_KNOBS:
        os.environ.pop(key, None)
    how = str(rng.choice(["default", "rounds", "rounds", "single"]))
    if how == "single":
        os.environ["RVLL_WALK_ROUNDS"] = "0"
    elif how == "rounds":
        os.environ["RVLL_WALK_ROUNDS"] = "1"
        os.environ["RVLL_ROUNDS_GROUPS"] = str(rng.integers(1, 5))
        if rng.random() < 0.5: os.environ["RVLL_ROUNDS_FREE"] = str(int(rng.choice([1, 300, 3000, 100000])))
        if rng.random() < 0.3: os.environ["RVLL_ROUNDS_DEPTH"] = str(rng.integers(1, 10))
        if rng.random() < 0.3: os.environ["RVLL_ROUNDS_FORM"] = "cu"
        if rng.random() < 0.3: os.environ["RVLL_ROUNDS_PB"] = str(rng.integers(1, 17))
        if rng.random() < 0.3: os.environ["RVLL_ROUNDS_W"] = str(int(rng.choice([8, 16, 32, 64])))
    nst, mr, sd = int(rng.integers(1, 30)), int(rng.choice([1, 3, 200])), int(rng.integers(0, 2 ** 62))
    c2, t2, l2, n = m.slice_walk(cube, theta, logl, lstar, chol, wrapped_params(m.parnames), nsteps=nst, max_rounds=mr, seed=sd)
    took_rounds = m.slice_walk_rounds() > 0
    hows[how + (" (rounds)" if took_rounds else " (single kernel)")] = hows.get(how + (" (rounds)" if took_rounds else " (single kernel)"), 0) + 1
    if rng.random() < 0.3:
        # the other form, same seed: end points, theta, log-L and the call count bit for bit
        for key in ROUNDS_KNOBS:
            os.environ.pop(key, None)
        os.environ["RVLL_WALK_ROUNDS"] = "0" if took_rounds else "1"
        c4, t4, l4, n4 = m.slice_walk(cube, theta, logl, lstar, chol, wrapped_params(m.parnames), nsteps=nst, max_rounds=mr, seed=sd)
        assert n4 == n and np.array_equal(c4, c2) and np.array_equal(t4, t2) and np.array_equal(l4, l2), "the two forms of the walk disagree"
        crosses += 1
        os.environ.pop("RVLL_WALK_ROUNDS", None)
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
        if rng.random() < 0.5:
            new, used = m.live_step(order, kd, start, ls, wrapped_params(m.parnames), nsteps=nst, seed=seed2, chol=chol)
        else:                                                # round 4: the order made on the device, start rows as ranks among the survivors
            dl, ls_dev, top = m.live_sort(kd)
            assert ls_dev == ls and np.array_equal(dl, ll0[order[:kd]]) and top == ll0.max()
            pos = np.empty(nrow, dtype=np.int64); pos[order] = np.arange(nrow)
            new, used = m.live_step(None, kd, (pos[start] - kd).astype(np.int32), ls, wrapped_params(m.parnames), nsteps=nst, seed=seed2, chol=chol)
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
      f"{lives} repeated through the resident live set; forms taken {dict(sorted(hows.items()))}, {crosses} walks repeated in the other form), "
      f"{calls} likelihood calls inside walks, {scalars} scalar-server calls, all consistent")
