"""Loaders for the FIP periodogram fixtures (tests/golden/fip_*.npz, written by the reference script)."""
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"
CASES = ("small", "edges", "single")


class FipCase:
    def __init__(self, name):
        z = np.load(GOLDEN / f"fip_{name}.npz")
        self.name = name
        self.nmod, self.reps = (int(v) for v in z["meta"])
        self.pmin, self.pmax = (float(v) for v in z["prange"])
        self.times = z["times"]
        self.tobs = float(self.times.max() - self.times.min())
        self.nu, self.fapnu = z["nu"], z["fapnu"]
        self.logzs = [[float(z[f"logZ_{r}_{k}"]) for k in range(self.nmod)] for r in range(self.reps)]
        self.posteriors = [[None] + [(z[f"samples_{r}_{k}"], z[f"weights_{r}_{k}"]) for k in range(1, self.nmod)]
                           for r in range(self.reps)]
