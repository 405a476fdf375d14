"""Synthetic workloads: the five BASELINE.json configurations as deterministic
datasets + parameter sets.  Used by bench.py, the tests and the
golden-vector generator; there is no model evaluation in this file.

  cfg  planets  epochs  instruments  drift  D   live points
   1   1 (e=0 fixed)     50   1   -    6     400
   2   1 (e=0.3 fixed)  200   1   -    6    4096
   3   3                200   2   -   19   16384
   4   3               1000   2   -   19   65536  (8192 x 8 GPUs)
   5   5               2000   3  lin  32  131072  (16384 x 8 GPUs)

Parameter families are the ones the reference's shipped configs use
(evidence/examples/51Peg/config_51Peg_example.py:43-53, config_51Peg_drift.py:56-58).
"""
from dataclasses import dataclass
from typing import Dict, List

import numpy as np

from . import priors as P
from .data import EpochTable

CONFIGS = {
    1: dict(nplanets=1, n_epochs=50, ninst=1, drift=False, ecc_fixed=0.0, batch=400),
    2: dict(nplanets=1, n_epochs=200, ninst=1, drift=False, ecc_fixed=0.3, batch=4096),
    3: dict(nplanets=3, n_epochs=200, ninst=2, drift=False, ecc_fixed=None, batch=16384),
    4: dict(nplanets=3, n_epochs=1000, ninst=2, drift=False, ecc_fixed=None, batch=65536),
    5: dict(nplanets=5, n_epochs=2000, ninst=3, drift=True, ecc_fixed=None, batch=131072),
}
INST_NAMES = ["harps", "hires", "espresso"]
EPOCH0 = 51000.0
TWO_PI = 2.0 * np.pi


@dataclass
class Workload:
    cfg: int
    table: EpochTable
    parnames: List[str]                 # sorted free names
    fixedpardict: Dict[str, float]
    input_dict: dict                    # reference config format [value, flag, [Prior, *args]]
    batch: int

    @property
    def ndim(self):
        return len(self.parnames)

    def priordict(self):
        return P.prior_constructor(self.input_dict)

    def sample_theta(self, n, seed):
        """n live points drawn from the priors (direct sampling, no quantile functions)."""
        rng = np.random.default_rng(seed)
        cols = []
        for name in self.parnames:
            obj, par = name.rsplit("_", 1)
            prior = self.input_dict[obj][par][2]
            kind, args = prior[0], prior[1:]
            u = rng.random(n)
            if kind == "Uniform":
                cols.append(args[0] + (args[1] - args[0]) * u)
            elif kind == "Jeffreys":
                cols.append(args[0] * (args[1] / args[0]) ** u)
            elif kind == "UniformFrequency":
                cols.append(args[0] / (1 - u * (args[1] - args[0]) / args[1]))
            elif kind == "Beta":
                cols.append(rng.beta(args[0], args[1], size=n))
            else:
                raise ValueError(kind)
        return np.ascontiguousarray(np.stack(cols, axis=1))

    def sample_cube(self, n, seed):
        return np.random.default_rng(seed).random((n, self.ndim))


def make_workload(cfg: int) -> Workload:
    spec = CONFIGS[cfg]
    rng = np.random.default_rng(1000 + cfg)
    ne, ni, npl = spec["n_epochs"], spec["ninst"], spec["nplanets"]
    t = np.sort(rng.uniform(50000.0, 52000.0, ne))
    inst = rng.integers(0, ni, ne)
    svrad = rng.uniform(0.5, 3.0, ne)
    # planted signal: circular-orbit sinusoids + per-instrument offsets (+ slope), white noise
    truth_k = [12.0, 5.0, 3.0, 2.0, 1.5][:npl]
    truth_p = [17.3, 61.0, 143.0, 7.7, 402.0][:npl]
    truth_phi = rng.uniform(0, TWO_PI, npl)
    signal = np.zeros(ne)
    for k, p, phi in zip(truth_k, truth_p, truth_phi):
        signal += k * np.cos(TWO_PI * (t - EPOCH0) / p + phi)
    signal += np.array([1.5, -2.0, 0.7])[inst]
    if spec["drift"]:
        signal += 3.0 * (t - EPOCH0) / 365.25
    vrad = signal + rng.normal(0.0, np.sqrt(svrad ** 2 + 1.0))
    # concatenate instrument by instrument, like evidence/rvmodel/__init__.py:50-55
    order = np.argsort(inst, kind="stable")
    names = INST_NAMES[:ni]
    table = EpochTable.from_arrays(names, t[order], vrad[order], svrad[order], inst[order].astype(np.int32))

    input_dict = {}
    for n in range(1, npl + 1):
        planet = {"k1": [0.0, 1, ["Jeffreys", 0.1, 100.0]],
                  "period": [0.0, 1, ["UniformFrequency", 1.5, 1000.0]],
                  "omega": [0.1, 1, ["Uniform", 0.0, TWO_PI]],
                  "ma0": [0.1, 1, ["Uniform", 0.0, TWO_PI]],
                  "epoch": [EPOCH0, 0]}
        if spec["ecc_fixed"] is None:
            planet["ecc"] = [0.1, 1, ["Beta", 0.867, 3.03]]
        else:
            planet["ecc"] = [spec["ecc_fixed"], 0]
        input_dict[f"planet{n}"] = planet
    for name in names:
        input_dict[name] = {"offset": [0.0, 1, ["Uniform", -10.0, 10.0]],
                            "jitter": [0.75, 1, ["Uniform", 0.0, 50.0]]}
    if spec["drift"]:
        input_dict["drift"] = {"lin": [0.0, 1, ["Uniform", -100.0, 100.0]], "tref": [EPOCH0, 0]}

    parnames, fixed = [], {}
    for obj, pars in input_dict.items():
        for par, entry in pars.items():
            if entry[1] == 0:
                fixed[f"{obj}_{par}"] = float(entry[0])
            else:
                parnames.append(f"{obj}_{par}")
    return Workload(cfg, table, sorted(parnames), fixed, input_dict, spec["batch"])
