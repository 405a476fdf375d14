"""Epoch table: the concatenated RV time series the kernels keep resident in HBM.

Mirrors evidence/rvmodel/__init__.py:46-55 (instruments concatenated in datadict
order, an integer instrument id per epoch — NOT time-sorted) and :141-146 (time
column `rjd`, else `jdb`; `vrad`; `svrad`).
"""
from dataclasses import dataclass
from typing import List

import numpy as np


def _column(table, key):
    """Column `key` of a pandas DataFrame, a dict of arrays or a numpy record array."""
    try:
        col = table[key]
    except (KeyError, ValueError, IndexError):
        return None
    return np.asarray(getattr(col, "values", col), dtype=np.float64)


@dataclass
class EpochTable:
    insts: List[str]          # instrument names, datadict order
    time: np.ndarray          # [Ne] float64
    vrad: np.ndarray          # [Ne] float64
    svrad: np.ndarray         # [Ne] float64
    inst_id: np.ndarray       # [Ne] int32

    @property
    def n_epochs(self):
        return int(self.time.shape[0])

    @classmethod
    def from_datadict(cls, datadict):
        """datadict: {instrument: {'data': table}} as built by evidence/config.py:102-114."""
        insts = list(datadict.keys())
        if not insts:
            raise ValueError("datadict holds no instrument")
        t, y, s, ids = [], [], [], []
        time_keys = set()
        for i, name in enumerate(insts):
            table = datadict[name]["data"]
            tt = _column(table, "rjd")
            key = "rjd"
            if tt is None:
                tt, key = _column(table, "jdb"), "jdb"
            if tt is None:
                raise KeyError(f"instrument {name!r}: neither 'rjd' nor 'jdb' column")
            time_keys.add(key)
            vv, ss = _column(table, "vrad"), _column(table, "svrad")
            if vv is None or ss is None:
                raise KeyError(f"instrument {name!r}: 'vrad'/'svrad' column missing")
            if not (len(tt) == len(vv) == len(ss)):
                raise ValueError(f"instrument {name!r}: column lengths differ")
            t.append(tt); y.append(vv); s.append(ss)
            ids.append(np.full(len(tt), i, dtype=np.int32))
        if len(time_keys) > 1:
            raise ValueError("instruments mix 'rjd' and 'jdb' time columns")
        return cls(insts, np.ascontiguousarray(np.concatenate(t)), np.ascontiguousarray(np.concatenate(y)),
                   np.ascontiguousarray(np.concatenate(s)), np.ascontiguousarray(np.concatenate(ids)))

    @classmethod
    def from_arrays(cls, insts, time, vrad, svrad, inst_id):
        time = np.ascontiguousarray(time, dtype=np.float64)
        vrad = np.ascontiguousarray(vrad, dtype=np.float64)
        svrad = np.ascontiguousarray(svrad, dtype=np.float64)
        inst_id = np.ascontiguousarray(inst_id, dtype=np.int32)
        if not (time.shape == vrad.shape == svrad.shape == inst_id.shape) or time.ndim != 1:
            raise ValueError("time, vrad, svrad, inst_id must be 1-D arrays of one length")
        if time.size == 0:
            raise ValueError("empty epoch table")
        if inst_id.min() < 0 or inst_id.max() >= len(insts):
            raise ValueError("inst_id outside [0, number of instruments)")
        return cls(list(insts), time, vrad, svrad, inst_id)

    def to_datadict(self):
        """The reference's datadict shape ({inst: {'data': {...columns...}}}), one entry per instrument."""
        out = {}
        for i, name in enumerate(self.insts):
            m = self.inst_id == i
            out[name] = {"data": {"rjd": self.time[m].copy(), "vrad": self.vrad[m].copy(),
                                  "svrad": self.svrad[m].copy()}}
        return out
