#!/usr/bin/env python3
"""Where do the device's Newton iterates leave the reference's?  For the worst rows of the eccentricity-sweep fixture
(tests/golden/loglike_high_ecc.npz) the solver is replayed on the host twice — sin / cos from numpy, and from the device's
SHORT-reduction routine (rvll_debug_eval ops 0 / 1), everything else in IEEE double as both sides do it — epoch by epoch.
This is the probe that found the excursions to |E| ~ 1e9 .. 1e22 at the 0.99 clamp: beyond 2^51 * pi/2 the
short reduction returns sin / cos of another angle.  (The kernels now switch to the long reduction there; op 0 / 1 of
rvll_debug_eval still show the short one.)  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import golden
from evidence_amd import GpuRVModel
from oracle import oracle as orc

case = golden.high_ecc_case()
names = case.parnames
with GpuRVModel(case.fixed, case.table, names) as m:
    got = m.log_likelihood_batch(case.theta)
    err = golden.rel_err(got, case.logL)
    worst = np.argsort(-err)[:4]
    print("worst rows:", [(int(i), float(err[i])) for i in worst])
    t = case.table.time
    for row in worst[:2]:
        th = dict(zip(names, case.theta[row]))
        e = min(th["planet1_ecc"], 0.99)
        M = 2 * np.pi / th["planet1_period"] * (t - case.fixed["planet1_epoch"]) + th["planet1_ma0"]
        E_a, E_b = M.copy(), M.copy()
        act_a = np.ones(M.size, bool); act_b = act_a.copy()
        first_div = np.full(M.size, -1)
        for it in range(3000):
            if not (act_a.any() or act_b.any()):
                break
            sa, ca = np.sin(E_a), np.cos(E_a)
            sb, cb = m.debug_eval(0, E_b), m.debug_eval(1, E_b)
            na = E_a - (E_a - e * sa - M) / (1 - e * ca)
            nb = E_b - (E_b - e * sb - M) / (1 - e * cb)
            da, db = np.abs(na - E_a), np.abs(nb - E_b)
            E_a = np.where(act_a, na, E_a); E_b = np.where(act_b, nb, E_b)
            newly = (first_div < 0) & (E_a != E_b)
            first_div[newly] = it
            act_a &= da > 1e-4; act_b &= db > 1e-4
        dE = np.abs(E_a - E_b)
        j = int(np.argmax(dE))
        print(f"row {row}: e = {th['planet1_ecc']:.4f}; epochs whose iterates ever differ {int((first_div >= 0).sum())} of {M.size}; "
              f"final |E_libm - E_device| max {dE.max():.3e} at epoch {j} (first difference at step {first_div[j]}), "
              f"epochs with final difference > 1e-9: {int((dE > 1e-9).sum())}")
        # at the first differing step of that epoch: how far apart are the two sin / cos?
        Ea = M[j].copy()
        for it in range(first_div[j] + 1):
            s_l, c_l = np.sin(Ea), np.cos(Ea)
            s_d, c_d = m.debug_eval(0, np.array([Ea]))[0], m.debug_eval(1, np.array([Ea]))[0]
            if it == first_div[j]:
                print(f"    at step {it}: E = {Ea!r}; sin libm {s_l!r} device {s_d!r} ({(s_d - s_l) / np.spacing(abs(s_l)):+.1f} ulp); "
                      f"cos libm {c_l!r} device {c_d!r} ({(c_d - c_l) / np.spacing(abs(c_l)):+.1f} ulp); f' = {1 - e * c_l:.3e}")
            Ea = Ea - (Ea - e * s_l - M[j]) / (1 - e * c_l)
