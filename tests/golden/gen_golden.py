#!/usr/bin/env python3
"""Golden-vector generator — runs ONLY in the build container, where the reference
checkout is mounted at /root/reference.  It imports the reference's own
`evidence.rvmodel.RVModel` and `evidence.priors`, evaluates them on fixed inputs and
writes small fixtures (inputs + expected outputs) next to this file:

    loglike_cfg{1..5}.npz   synthetic BASELINE.json configurations
    loglike_edges.npz/.json parametrisation and edge cases
    loglike_51peg.npz/.json the shipped 51Peg example, both configs
    priors.npz/.json        .ppf of every working distribution on a q grid (§8c.4)
    keprv.npz/.json         kep_rv(exclude_planet) / modelk(planet) curves at arbitrary times (§8f.3)
    loglike_high_ecc.npz    eccentricity sweep 0.90 .. 0.9925 (the solver's sensitive corner)
    loglike_wild.npz        adversarial parameter ranges (tiny / huge periods and amplitudes, invalid orbits, zero jitter)

The reference never travels to the GPU box; these fixtures do.  Dev-only shim:
`numpy.int = int` (evidence/rvmodel/__init__.py:53 uses the alias numpy removed).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py
"""
import json
import sys
import warnings
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
np.int = int  # noqa: dev-only shim for the reference under numpy 2.x

import pandas as pd  # noqa: E402
from evidence import priors as ref_priors  # noqa: E402
from evidence.rvmodel import RVModel  # noqa: E402

from evidence_amd.synthetic import make_workload, EPOCH0  # noqa: E402
from evidence_amd.data import EpochTable  # noqa: E402

warnings.filterwarnings("ignore")


def ref_datadict(table: EpochTable, time_key="rjd"):
    out = {}
    for i, name in enumerate(table.insts):
        m = table.inst_id == i
        out[name] = {"data": pd.DataFrame({time_key: table.time[m], "vrad": table.vrad[m],
                                           "svrad": table.svrad[m]})}
    return out


def ref_loglike(table, parnames, fixed, thetas, time_key="rjd", linpar=None):
    model = RVModel(dict(fixed), ref_datadict(table, time_key), list(parnames))
    assert model.parnames == sorted(parnames)
    if linpar:
        model.linpar_dict = {k: np.asarray(v, dtype=float) for k, v in linpar.items()}
    return np.array([float(model.log_likelihood(np.asarray(x))) for x in thetas])


def table_arrays(prefix, table):
    return {f"{prefix}time": table.time, f"{prefix}vrad": table.vrad, f"{prefix}svrad": table.svrad,
            f"{prefix}inst_id": table.inst_id}


# ---------------------------------------------------------------------------------------
def gen_configs():
    for cfg in (1, 2, 3, 4, 5):
        w = make_workload(cfg)
        n = {1: 96, 2: 96, 3: 96, 4: 48, 5: 32}[cfg]
        theta = w.sample_theta(n, seed=4000 + cfg)
        logl = ref_loglike(w.table, w.parnames, w.fixedpardict, theta)
        np.savez_compressed(HERE / f"loglike_cfg{cfg}.npz", theta=theta, logL=logl,
                            parnames=np.array(w.parnames), insts=np.array(w.table.insts),
                            fixed_names=np.array(list(w.fixedpardict)),
                            fixed_values=np.array(list(w.fixedpardict.values()), dtype=float),
                            **table_arrays("", w.table))
        print(f"cfg{cfg}: {n} points, D={w.ndim}, logL in [{logl.min():.3f}, {logl.max():.3f}]")


def small_table(seed, n_epochs, ninst):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(50000.0, 50400.0, n_epochs))
    inst = rng.integers(0, ninst, n_epochs)
    inst[:ninst] = np.arange(ninst)          # every instrument present
    sv = rng.uniform(0.5, 3.0, n_epochs)
    y = 8.0 * np.cos(2 * np.pi * (t - 50000.0) / 23.0) + rng.normal(0, 2.0, n_epochs)
    order = np.argsort(inst, kind="stable")
    return EpochTable.from_arrays(["ia", "ib", "ic"][:ninst], t[order], y[order], sv[order],
                                  inst[order].astype(np.int32))


def gen_edges():
    rng = np.random.default_rng(77)
    cases = []
    arrays = {}
    tables = {"t1": small_table(11, 64, 1), "t2": small_table(12, 90, 2), "t3": small_table(13, 130, 3)}
    for key, tb in tables.items():
        arrays.update(table_arrays(f"{key}_", tb))

    def add(name, table_key, free, fixed, theta, time_key="rjd", linpar=None, note=""):
        free = sorted(free)
        theta = np.atleast_2d(np.asarray(theta, dtype=float))
        assert theta.shape[1] == len(free), (name, theta.shape, len(free))
        logl = ref_loglike(tables[table_key], free, fixed, theta, time_key, linpar)
        i = len(cases)
        arrays[f"c{i}_theta"] = theta
        arrays[f"c{i}_logL"] = logl
        if linpar:
            for k, v in linpar.items():
                arrays[f"c{i}_linpar_{k}"] = np.asarray(v, dtype=float)
        cases.append(dict(name=name, table=table_key, parnames=free, fixed=fixed, time_key=time_key,
                          linpar=sorted(linpar) if linpar else [], note=note))
        print(f"edge {i:2d} {name:34s} n={len(logl):3d} logL[0]={logl[0]!r}")

    def draw(free, n, ecc=None):
        """Random theta over sorted(free) with sensible ranges keyed by the parameter suffix."""
        cols = []
        for nm in sorted(free):
            suf = nm.split("_", 1)[1]
            u = rng.random(n)
            if suf == "k1": c = 0.5 + 30 * u
            elif suf == "logk1": c = np.log(0.5 + 30 * u)
            elif suf == "period": c = 2.0 + 200 * u
            elif suf == "logperiod": c = np.log(2.0 + 200 * u)
            elif suf == "ecc": c = (rng.beta(0.867, 3.03, n) if ecc is None else np.full(n, ecc))
            elif suf in ("omega", "ma0", "ml0"): c = 2 * np.pi * u
            elif suf in ("secos", "sesin"): c = rng.uniform(-0.65, 0.65, n)
            elif suf in ("ecos", "esin"): c = rng.uniform(-0.6, 0.6, n)
            elif suf == "offset": c = -10 + 20 * u
            elif suf == "jitter": c = 10 * u
            elif suf in ("lin", "quad", "cub", "quar"): c = rng.uniform(-5, 5, n)
            elif suf == "tref": c = 50100 + 100 * u
            elif suf == "epoch": c = 50000 + 50 * u
            elif suf.startswith("rhk") or suf.startswith("fwhm"): c = rng.uniform(-3, 3, n)
            else: raise KeyError(nm)
            cols.append(c)
        return np.stack(cols, axis=1)

    base1 = ["planet1_k1", "planet1_period", "planet1_ecc", "planet1_omega", "planet1_ma0",
             "ia_offset", "ia_jitter"]
    fx1 = {"planet1_epoch": 50000.0}
    add("direct_ecc_random", "t1", base1, fx1, draw(base1, 24))
    add("ecc_zero", "t1", base1, fx1, draw(base1, 8, ecc=0.0))
    add("ecc_0p3", "t1", base1, fx1, draw(base1, 8, ecc=0.3))
    add("ecc_0p6", "t1", base1, fx1, draw(base1, 8, ecc=0.6))
    add("ecc_0p9", "t1", base1, fx1, draw(base1, 8, ecc=0.9))
    add("ecc_0p95", "t1", base1, fx1, draw(base1, 8, ecc=0.95))
    add("ecc_clamp_below_0p989", "t1", base1, fx1, draw(base1, 4, ecc=0.989),
        note="just below the 0.99 clamp of trueanomaly.c:11-12")
    add("ecc_above_clamp_0p995", "t1", base1, fx1, draw(base1, 4, ecc=0.995),
        note="solver clamps to 0.99, RV formula keeps 0.995 (rvmodel:463)")
    add("ecc_above_one_direct", "t1", base1, fx1, draw(base1, 4, ecc=1.2),
        note="direct parametrisation has no ecc>1 guard (rvmodel:441-447)")
    add("ecc_negative_direct", "t1", base1, fx1, draw(base1, 4, ecc=-0.2))

    f = ["planet1_k1", "planet1_period", "planet1_secos", "planet1_sesin", "planet1_ml0", "ia_offset", "ia_jitter"]
    th = draw(f, 24)
    add("secos_sesin_ml0", "t1", f, fx1, th)
    th2 = draw(f, 6)
    idx = sorted(f)
    th2[:, idx.index("planet1_secos")] = 0.9
    th2[:, idx.index("planet1_sesin")] = 0.8
    add("secos_sesin_invalid", "t1", f, fx1, th2, note="secos^2+sesin^2 > 1 -> -1e30 (rvmodel:430-431,203)")

    f = ["planet1_k1", "planet1_period", "planet1_ecos", "planet1_esin", "planet1_ma0", "ia_offset", "ia_jitter"]
    add("ecos_esin_ma0", "t1", f, fx1, draw(f, 24))
    th2 = draw(f, 6)
    idx = sorted(f)
    th2[:, idx.index("planet1_ecos")] = 0.9
    th2[:, idx.index("planet1_esin")] = 0.7
    add("ecos_esin_invalid", "t1", f, fx1, th2, note="sqrt(ecos^2+esin^2) > 1 -> -1e30 (rvmodel:438-439)")

    f = ["planet1_logk1", "planet1_logperiod", "planet1_ecc", "planet1_omega", "planet1_ml0", "ia_offset", "ia_jitter"]
    add("logk1_logperiod_ml0", "t1", f, fx1, draw(f, 16))

    f = ["planet1_k1", "planet1_period", "planet1_ecc", "planet1_omega", "planet1_ma0", "planet1_epoch",
         "ia_offset", "ia_jitter"]
    add("free_epoch", "t1", f, {}, draw(f, 12))

    f = ["planet1_k1", "planet1_period", "planet1_ecc", "planet1_omega", "planet1_ma0", "ia_offset"]
    add("no_jitter", "t1", f, fx1, draw(f, 12), note="no free name contains 'jitter' (rvmodel:138-139,191-192)")
    add("no_jitter_fixed_jitter_ignored", "t1", f, dict(fx1, ia_jitter=4.0), draw(f, 6),
        note="a FIXED jitter does not switch jitter on (flags come from free names only)")

    f = ["planet1_k1", "planet1_period", "planet1_omega", "planet1_ma0", "ia_offset", "ia_jitter"]
    add("fixed_ecc_omega_mix", "t1", f, dict(fx1, planet1_ecc=0.41), draw(f, 8))

    f = ["ia_offset", "ia_jitter"]
    add("zero_planets", "t1", f, {}, draw(f, 8), note="nplanets = 0 (rvmodel:195)")
    add("fixed_k1_is_not_a_planet", "t1", ["ia_offset", "ia_jitter", "planet1_period"],
        dict(fx1, planet1_k1=5.0, planet1_ecc=0.1, planet1_omega=1.0, planet1_ma0=2.0),
        draw(["ia_offset", "ia_jitter", "planet1_period"], 4),
        note="planets are counted over FREE names containing 'k1' (rvmodel:122-124)")

    two = []
    for n in (1, 2):
        two += [f"planet{n}_k1", f"planet{n}_period", f"planet{n}_ecc", f"planet{n}_omega", f"planet{n}_ma0"]
    fx2 = {"planet1_epoch": 50000.0, "planet2_epoch": 50010.0}
    f = two + ["ia_offset", "ia_jitter", "ib_offset", "ib_jitter"]
    add("two_planets_two_inst", "t2", f, fx2, draw(f, 24))

    for order, keys in ((1, ["lin"]), (2, ["lin", "quad"]), (3, ["lin", "quad", "cub"]),
                        (4, ["lin", "quad", "cub", "quar"])):
        f = base1 + [f"drift_{k}" for k in keys]
        add(f"drift_order{order}_tref_fixed", "t1", f, dict(fx1, drift_tref=50200.0), draw(f, 10))
        add(f"drift_order{order}_tref_data", "t1", f, fx1, draw(f, 10), note="tref = time[0] (rvmodel:259-260)")
    f = base1 + ["drift_lin", "drift_tref"]
    add("drift_free_tref", "t1", f, fx1, draw(f, 10))
    f = base1 + ["drift_quad"]
    add("drift_quad_only_fixed_lin", "t1", f, dict(fx1, drift_lin=1.25), draw(f, 8))

    f3 = two + ["ia_offset", "ia_jitter", "ib_offset", "ib_jitter", "ic_offset", "ic_jitter", "drift_lin"]
    add("three_inst_drift_tref_data", "t3", f3, fx2, draw(f3, 16),
        note="time[0] is the first epoch of the FIRST instrument, not the earliest time")
    add("jdb_time_column", "t2", f, dict(fx1, ib_offset=0.5, ib_jitter=1.0), draw(f, 8), time_key="jdb")

    # linear activity terms (rvmodel:210-212)
    rng2 = np.random.default_rng(5)
    t1n = tables["t1"].n_epochs
    lin = {"rhk": rng2.normal(0, 1, t1n), "fwhm": rng2.normal(0, 2, t1n)}
    f = base1 + ["linpar_rhk", "linpar_fwhm"]
    add("linpar_two_series", "t1", f, fx1, draw(f, 12), linpar=lin)

    # high eccentricity near the clamp with many iterations (no itmax hit expected)
    add("ecc_0p98", "t1", base1, fx1, draw(base1, 6, ecc=0.98))

    np.savez_compressed(HERE / "loglike_edges.npz", **arrays)
    (HERE / "loglike_edges.json").write_text(json.dumps(cases, indent=1))


def gen_51peg():
    df = pd.read_csv("/root/reference/evidence/examples/51Peg/51Peg.rv", sep="\t", skiprows=(1,))
    table = EpochTable.from_arrays(["hamilton"], df["rjd"].values, df["vrad"].values, df["svrad"].values,
                                   np.zeros(len(df), dtype=np.int32))
    rng = np.random.default_rng(51)
    out = {}
    meta = []
    base = ["hamilton_jitter", "hamilton_offset", "planet1_ecc", "planet1_k1", "planet1_ma0", "planet1_omega",
            "planet1_period"]

    def thetas(names, n):
        cols = []
        for nm in names:
            u = rng.random(n)
            cols.append({"hamilton_jitter": 50 * u, "hamilton_offset": -10 + 20 * u,
                         "planet1_ecc": rng.beta(0.867, 3.03, n), "planet1_k1": 0.1 * 1000 ** u,
                         "planet1_ma0": 2 * np.pi * u, "planet1_omega": 2 * np.pi * u,
                         "planet1_period": 1 / (1 - u * 99 / 100), "drift_lin": -100 + 200 * u}[nm])
        return np.stack(cols, axis=1)

    th = thetas(base, 63)
    th = np.vstack([[3.0, -2.0, 0.05, 56.0, 1.0, 0.7, 4.2308], th])     # BASELINE.md known answer first
    fixed = {"planet1_epoch": 51050.0}
    ll = ref_loglike(table, base, fixed, th)
    assert repr(float(ll[0])) == "-11539.57252446112", ll[0]
    out["example_theta"], out["example_logL"] = th, ll
    meta.append(dict(name="example", parnames=base, fixed=fixed))
    names = sorted(base + ["drift_lin"])
    fixed2 = {"planet1_epoch": 51050.0, "drift_tref": 51050.0}
    th = thetas(names, 64)
    out["drift_theta"], out["drift_logL"] = th, ref_loglike(table, names, fixed2, th)
    meta.append(dict(name="drift", parnames=names, fixed=fixed2))
    np.savez_compressed(HERE / "loglike_51peg.npz", **table_arrays("", table), **out)
    (HERE / "loglike_51peg.json").write_text(json.dumps(meta, indent=1))
    print("51Peg: known answer", repr(float(ll[0])))


def gen_priors():
    rng = np.random.default_rng(9)
    q = np.concatenate([[0.0, 1e-300, 1e-12, 1e-6, 1e-3, 0.01, 0.1, 0.25, 0.37, 0.5, 0.63, 0.75, 0.9, 0.99, 0.999,
                         1 - 1e-6, 1 - 1e-12, 1.0], rng.random(46),
                        # round 2 (appended, so the indices above stay): the far tails, so that what the tests exclude
                        # for the special-function kinds is a statement about fixtures, not a mask
                        [1e-15, 1e-14, 1e-13, 1e-9, 1 - 1e-7, 1 - 1e-8, 1 - 1e-9, 1 - 1e-10, 1 - 1e-13, 1 - 1e-15]])
    sets = [
        ("Uniform", (4, 6)), ("Uniform", (-10, 10)), ("Uniform", (0.0, 2 * np.pi)),
        ("Jeffreys", (10, 100)), ("Jeffreys", (0.1, 100.0)),
        ("ModJeffreys", (1.0, 100.0)), ("ModJeffreys", (0.5, 2000.0)),
        ("UniformFrequency", (1, 100)), ("UniformFrequency", (1.5, 1000.0)), ("UniformFrequency", (1, 1e3)),
        ("Normal", (0.0, 1.0)), ("Normal", (3.5, 0.25)),
        ("LogNormal", (0.5,)), ("LogNormal", (1.2, 0.0, 3.0)),
        ("TruncatedRayleigh", (0.2, 1.0)), ("TruncatedRayleigh", (2.0, 5.0)),
        ("Binormal", (0.0, 1.0, 4.0, 0.5, 0.3)), ("Binormal", (-2.0, 0.3, -1.0, 2.0, -0.6)),
        ("AsymmetricNormal", (1.0, 0.5, 2.0)), ("AsymmetricNormal", (-3.0, 2.0, 0.1)),
        ("TruncatedUNormal", (0.0, 1.0, -1.0, 2.0)), ("TruncatedUNormal", (5.0, 2.0, 0.0, 6.0)),
        ("PowerLaw", (-2.0, 1.0, 10.0)), ("PowerLaw", (0.5, 0.0, 3.0)),
        ("DoublePowerLaw", (-0.5, -2.5, 3.0, 1.0, 30.0)), ("DoublePowerLaw", (1.0, -1.5, 0.5, 0.0, 4.0)),
        ("Sine", (0.0, 180.0)), ("Sine", (20.0, 90.0)),
        ("Alpha", (1.5,)), ("Alpha", (3.0,)),
        ("Beta", (0.867, 3.03)), ("Beta", (2.0, 5.0)), ("Beta", (0.5, 0.5)), ("Beta", (12.0, 1.5)),
        ("Gamma", (2.0, 3.0)), ("Gamma", (0.7, 0.1)), ("Gamma", (25.0, 2.0)),
        # round 2, second pass (appended): shape parameters far from anything a config would use
        ("Beta", (0.05, 0.05)), ("Beta", (0.1, 20.0)), ("Beta", (30.0, 0.3)), ("Beta", (150.0, 300.0)), ("Beta", (1.0, 1.0)),
        ("Beta", (1.0, 7.0)), ("Beta", (500.0, 2.0)), ("Beta", (0.3, 0.31)),
        ("Gamma", (0.05, 1.0)), ("Gamma", (0.2, 5.0)), ("Gamma", (1.0, 0.5)), ("Gamma", (100.0, 0.01)), ("Gamma", (1000.0, 3.0)),
        ("Gamma", (5.5, 1e3)),
        ("Alpha", (0.3,)), ("Alpha", (10.0,)),
        ("Normal", (1e6, 1e-3)), ("LogNormal", (3.0,)), ("LogNormal", (0.05, 2.0, 0.1)),
        ("TruncatedRayleigh", (10.0, 1.0)), ("Jeffreys", (1e-8, 1e8)), ("ModJeffreys", (1e-3, 1e6)),
        ("UniformFrequency", (0.01, 1e5)),
    ]
    arrays, meta = {"q": q}, []
    for i, (name, args) in enumerate(sets):
        dist = getattr(ref_priors, name)(*args)
        vals = np.empty_like(q)
        raised = np.zeros(q.shape, dtype=bool)
        for k, qq in enumerate(q):
            try:
                vals[k] = float(np.asarray(dist.ppf(qq)).reshape(-1)[0])
            except Exception:
                vals[k], raised[k] = np.nan, True
        arrays[f"p{i}_ppf"], arrays[f"p{i}_raised"] = vals, raised
        meta.append(dict(name=name, args=[float(a) for a in args]))
        print(f"prior {i:2d} {name:18s}{str(args):34s} ppf(0.5)={vals[9]!r} raised={int(raised.sum())}")
    # Log10Normal is broken upstream (TypeError from numpy.linspace with a float count)
    try:
        ref_priors.Log10Normal(0.0, 1.0).ppf(0.5)
        log10_state = "works"
    except TypeError as exc:
        log10_state = f"TypeError: {exc}"
    # the reference's own unit tests, tests/test_priors.py:13-15
    u = ref_priors.Uniform(4, 6)
    assert u.ppf(0.5) == 5 and u.ppf(0) == 4 and u.ppf(1) == 6
    np.savez_compressed(HERE / "priors.npz", **arrays)
    (HERE / "priors.json").write_text(json.dumps(dict(sets=meta, log10normal_upstream=log10_state), indent=1))
    # the SURVEY's known prior(cube = 0.37) answer for the shipped 51Peg config
    cfg = [("hamilton_jitter", "Uniform", (0.0, 50.0)), ("hamilton_offset", "Uniform", (-10, 10)),
           ("planet1_ecc", "Beta", (0.867, 3.03)), ("planet1_k1", "Jeffreys", (0.1, 100.0)),
           ("planet1_ma0", "Uniform", (0.0, 2 * np.pi)), ("planet1_omega", "Uniform", (0.0, 2 * np.pi)),
           ("planet1_period", "UniformFrequency", (1, 100))]
    print("51Peg prior(0.37):", [float(getattr(ref_priors, n)(*a).ppf(0.37)) for _, n, a in cfg])


def gen_keprv():
    """kep_rv(exclude_planet) and modelk(planet) curves (evidence/rvmodel/__init__.py:343-463) at times that
    are NOT the data epochs, as post_processing.py:413-428 calls them."""
    arrays, meta = {}, []
    for cfg, n in ((3, 6), (2, 4)):
        w = make_workload(cfg)
        theta = w.sample_theta(n, seed=7000 + cfg)
        times = np.linspace(49990.0, 52010.0, 57)
        model = RVModel(dict(w.fixedpardict), ref_datadict(w.table), list(w.parnames))
        npl = model.nplanets
        curves = {}
        for k in range(n):
            pardict = {name: theta[k, i] for i, name in enumerate(model.parnames)}
            pardict.update(model.fixedpardict)
            for ex in [None] + list(range(1, npl + 1)):
                curves.setdefault(f"ex{ex}", []).append(model.kep_rv(pardict, times, exclude_planet=ex))
            for pl in range(1, npl + 1):
                curves.setdefault(f"pl{pl}", []).append(model.modelk(pardict, times, planet=pl))
        arrays[f"cfg{cfg}_theta"], arrays[f"cfg{cfg}_times"] = theta, times
        for key, rows in curves.items():
            arrays[f"cfg{cfg}_{key}"] = np.array(rows)
        meta.append(dict(cfg=cfg, nplanets=npl, keys=sorted(curves)))
        print(f"keprv cfg{cfg}: {n} points x {len(times)} times, {len(curves)} curve kinds")
    np.savez_compressed(HERE / "keprv.npz", **arrays)
    (HERE / "keprv.json").write_text(json.dumps(meta, indent=1))


def gen_high_ecc():
    """The solver's sensitive corner: Newton from E = M at eccentricities near the 0.99 clamp wanders before it settles
    (6 .. 1159 steps) and amplifies any difference in an iterate.  A sweep of 10 eccentricities x 24 points on
    a two-planet, two-instrument model (the second planet at a moderate eccentricity), log-L from the reference itself."""
    rng = np.random.default_rng(2024)
    table = small_table(21, 160, 2)
    free = sorted(["planet1_k1", "planet1_period", "planet1_ecc", "planet1_omega", "planet1_ma0",
                   "planet2_k1", "planet2_period", "planet2_ecc", "planet2_omega", "planet2_ma0",
                   "ia_offset", "ia_jitter", "ib_offset", "ib_jitter"])
    fixed = {"planet1_epoch": 50000.0, "planet2_epoch": 50007.0}
    eccs = [0.90, 0.93, 0.95, 0.965, 0.975, 0.98, 0.985, 0.989, 0.9899, 0.9925]
    rows = []
    for e in eccs:
        n = 24
        th = {"planet1_k1": rng.uniform(1, 40, n), "planet1_period": rng.uniform(2, 300, n), "planet1_ecc": np.full(n, e),
              "planet1_omega": rng.uniform(0, 2 * np.pi, n), "planet1_ma0": rng.uniform(0, 2 * np.pi, n),
              "planet2_k1": rng.uniform(1, 20, n), "planet2_period": rng.uniform(5, 120, n),
              "planet2_ecc": rng.beta(0.867, 3.03, n), "planet2_omega": rng.uniform(0, 2 * np.pi, n),
              "planet2_ma0": rng.uniform(0, 2 * np.pi, n), "ia_offset": rng.uniform(-5, 5, n), "ia_jitter": rng.uniform(0, 6, n),
              "ib_offset": rng.uniform(-5, 5, n), "ib_jitter": rng.uniform(0, 6, n)}
        rows.append(np.stack([th[k] for k in free], axis=1))
    theta = np.concatenate(rows)
    logl = ref_loglike(table, free, fixed, theta)
    np.savez_compressed(HERE / "loglike_high_ecc.npz", theta=theta, logL=logl, parnames=np.array(free),
                        insts=np.array(table.insts), fixed_names=np.array(list(fixed)),
                        fixed_values=np.array(list(fixed.values()), dtype=float), ecc_of_row=np.repeat(eccs, 24),
                        **table_arrays("", table))
    print(f"high ecc: {len(theta)} points, logL in [{logl.min():.3f}, {logl.max():.3f}]")


def gen_wild():
    """Adversarial sweep: parameters drawn from far wider ranges than any prior would allow — periods 0.01 .. 1e5 d
    (|M| up to 1e6 rad), amplitudes 1e-3 .. 1e4, (secos, sesin) over the whole unit disk and a little beyond (invalid
    orbits), directly parametrised eccentricities from -0.1 to 1.1, huge offsets and drifts, jitters down to zero —
    on a two-instrument model with one planet per parametrisation family; log-L from the reference itself."""
    rng = np.random.default_rng(31337)
    table = small_table(33, 120, 2)
    free = sorted(["planet1_logk1", "planet1_logperiod", "planet1_secos", "planet1_sesin", "planet1_ml0",
                   "planet2_k1", "planet2_period", "planet2_ecc", "planet2_omega", "planet2_ma0",
                   "ia_offset", "ia_jitter", "ib_offset", "ib_jitter", "drift_lin", "drift_quad"])
    fixed = {"planet1_epoch": 50000.0, "planet2_epoch": 50123.456}
    n = 400
    r = np.sqrt(rng.uniform(0, 1.1, n)); ang = rng.uniform(0, 2 * np.pi, n)
    th = {"planet1_logk1": rng.uniform(np.log(1e-3), np.log(1e4), n), "planet1_logperiod": rng.uniform(np.log(0.01), np.log(1e5), n),
          "planet1_secos": r * np.cos(ang), "planet1_sesin": r * np.sin(ang), "planet1_ml0": rng.uniform(-20, 20, n),
          "planet2_k1": rng.choice([-1, 1], n) * 10 ** rng.uniform(-3, 4, n), "planet2_period": 10 ** rng.uniform(-2, 5, n),
          "planet2_ecc": rng.uniform(-0.1, 1.1, n), "planet2_omega": rng.uniform(-10, 10, n), "planet2_ma0": rng.uniform(-50, 50, n),
          "ia_offset": rng.normal(0, 1e3, n), "ia_jitter": np.where(rng.random(n) < 0.2, 0.0, 10 ** rng.uniform(-3, 2, n)),
          "ib_offset": rng.normal(0, 10, n), "ib_jitter": 10 ** rng.uniform(-3, 2, n),
          "drift_lin": rng.normal(0, 100, n), "drift_quad": rng.normal(0, 10, n)}
    theta = np.stack([th[k] for k in free], axis=1)
    logl = ref_loglike(table, free, fixed, theta)
    np.savez_compressed(HERE / "loglike_wild.npz", theta=theta, logL=logl, parnames=np.array(free),
                        insts=np.array(table.insts), fixed_names=np.array(list(fixed)),
                        fixed_values=np.array(list(fixed.values()), dtype=float), **table_arrays("", table))
    print(f"wild: {len(theta)} points, {int((logl == -1e30).sum())} invalid, finite logL in "
          f"[{logl[logl > -1e29].min():.3e}, {logl.max():.3e}], non-finite {int((~np.isfinite(logl)).sum())}")


if __name__ == "__main__":
    which = sys.argv[1:] or ["configs", "edges", "51peg", "priors", "keprv", "high_ecc", "wild"]
    if "wild" in which: gen_wild()
    if "high_ecc" in which: gen_high_ecc()
    if "configs" in which: gen_configs()
    if "edges" in which: gen_edges()
    if "51peg" in which: gen_51peg()
    if "priors" in which: gen_priors()
    if "keprv" in which: gen_keprv()
