"""Loaders for the committed golden fixtures (tests/golden/*.npz, made by gen_golden.py
from the reference itself)."""
import json
from pathlib import Path

import numpy as np

from evidence_amd.data import EpochTable
from evidence_amd.layout import compile_layout

GOLDEN = Path(__file__).resolve().parent / "golden"


class Case:
    def __init__(self, name, table, parnames, fixed, theta, logL, linpar=None, note=""):
        self.name, self.table, self.parnames, self.fixed = name, table, list(parnames), dict(fixed)
        self.theta, self.logL, self.linpar, self.note = theta, logL, linpar or {}, note
        self.layout = compile_layout(self.parnames, self.fixed, table.insts, list(self.linpar))

    @property
    def linpar_series(self):
        if not self.layout.linpar_names:
            return None
        return np.stack([self.linpar[k] for k in self.layout.linpar_names])

    def __repr__(self):
        return f"Case({self.name})"


def _table(z, prefix, insts):
    return EpochTable.from_arrays(insts, z[f"{prefix}time"], z[f"{prefix}vrad"], z[f"{prefix}svrad"],
                                  z[f"{prefix}inst_id"])


def config_case(cfg):
    z = np.load(GOLDEN / f"loglike_cfg{cfg}.npz")
    insts = [str(s) for s in z["insts"]]
    fixed = {str(k): float(v) for k, v in zip(z["fixed_names"], z["fixed_values"])}
    return Case(f"cfg{cfg}", _table(z, "", insts), [str(s) for s in z["parnames"]], fixed, z["theta"], z["logL"])


def edge_cases():
    z = np.load(GOLDEN / "loglike_edges.npz")
    meta = json.loads((GOLDEN / "loglike_edges.json").read_text())
    inst_names = {"t1": ["ia"], "t2": ["ia", "ib"], "t3": ["ia", "ib", "ic"]}
    out = []
    for i, m in enumerate(meta):
        table = _table(z, f"{m['table']}_", inst_names[m["table"]])
        linpar = {k: z[f"c{i}_linpar_{k}"] for k in m["linpar"]}
        # gen_golden passed linpar as a dict in insertion order rhk, fwhm; the reference iterates that dict
        if linpar:
            linpar = {k: linpar[k] for k in ("rhk", "fwhm") if k in linpar}
        out.append(Case(m["name"], table, m["parnames"], m["fixed"], z[f"c{i}_theta"], z[f"c{i}_logL"], linpar,
                        m.get("note", "")))
    return out


def peg51_cases():
    z = np.load(GOLDEN / "loglike_51peg.npz")
    meta = json.loads((GOLDEN / "loglike_51peg.json").read_text())
    table = _table(z, "", ["hamilton"])
    return [Case(f"51peg_{m['name']}", table, m["parnames"], m["fixed"], z[f"{m['name']}_theta"],
                 z[f"{m['name']}_logL"]) for m in meta]


def high_ecc_case():
    """Eccentricity sweep 0.90 .. 0.9925 of one planet (gen_golden.py, gen_high_ecc): the solver's sensitive corner."""
    z = np.load(GOLDEN / "loglike_high_ecc.npz")
    insts = [str(s) for s in z["insts"]]
    fixed = {str(k): float(v) for k, v in zip(z["fixed_names"], z["fixed_values"])}
    return Case("high_ecc_sweep", _table(z, "", insts), [str(s) for s in z["parnames"]], fixed, z["theta"], z["logL"])


def wild_case():
    """Adversarial parameter ranges (gen_golden.py, gen_wild)."""
    z = np.load(GOLDEN / "loglike_wild.npz")
    insts = [str(s) for s in z["insts"]]
    fixed = {str(k): float(v) for k, v in zip(z["fixed_names"], z["fixed_values"])}
    return Case("wild_sweep", _table(z, "", insts), [str(s) for s in z["parnames"]], fixed, z["theta"], z["logL"])


def all_loglike_cases():
    return [config_case(c) for c in (1, 2, 3, 4, 5)] + edge_cases() + peg51_cases() + [high_ecc_case(), wild_case()]


def prior_sets():
    z = np.load(GOLDEN / "priors.npz")
    meta = json.loads((GOLDEN / "priors.json").read_text())
    q = z["q"]
    return q, [(m["name"], m["args"], z[f"p{i}_ppf"], z[f"p{i}_raised"]) for i, m in enumerate(meta["sets"])]


def rel_err(a, b):
    """|a-b| / max(|b|, 1e-300), elementwise; exact equality (incl. -1e30 sentinels, inf) counts as 0."""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    with np.errstate(invalid="ignore", divide="ignore"):
        err = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
    err = np.where(a == b, 0.0, err)
    err = np.where(np.isnan(a) & np.isnan(b), 0.0, err)
    return err
