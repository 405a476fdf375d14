"""Layout compiler: parameter NAMES -> packed parameter layout for the kernels.

In the reference the model structure lives in the parameter names
(`planet{n}_k1`, `{inst}_offset`, `drift_lin`, ...): every log-L call rebuilds a
dict and looks names up with try/except (evidence/rvmodel/__init__.py:173-178,
412-456).  Here that resolution happens once, on the host, and yields a table of
slots — each either "free parameter i of theta" or "fixed constant".

Rules mirrored (evidence/rvmodel/__init__.py):
  :43       theta is ordered as sorted(parnames)
  :178      fixed parameters override free ones of the same name (dict.update)
  :122-124  nplanets = number of FREE names containing 'k1'
  :128-139  drift / linpar / jitter are in the model iff a FREE name contains the word
  :412-420  k1 before logk1, period before logperiod
  :425-447  secos/sesin, else ecos/esin, else ecc/omega
  :449-454  ml0 (then ma0 = ml0 - omega), else ma0
  :246-260  drift_{lin,quad,cub,quar} default 0; tref = drift_tref, else time[0]
  :187-190  {inst}_offset always, {inst}_jitter iff jitter is in the model
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

from . import _abi


@dataclass(frozen=True)
class SlotSpec:
    """Where a model scalar comes from: theta[index] (free) or a constant (fixed)."""
    index: int = -1
    value: float = 0.0

    @property
    def is_free(self):
        return self.index >= 0

    def to_c(self):
        return _abi.Slot.free(self.index) if self.is_free else _abi.Slot.fixed(self.value)


@dataclass(frozen=True)
class PlanetSpec:
    k_kind: int
    p_kind: int
    ecc_kind: int
    anom_kind: int
    k: SlotSpec
    p: SlotSpec
    e1: SlotSpec
    e2: SlotSpec
    anom: SlotSpec
    epoch: SlotSpec


@dataclass(frozen=True)
class InstSpec:
    name: str
    offset: SlotSpec
    jitter: SlotSpec


@dataclass
class ModelLayout:
    parnames: List[str]
    planets: List[PlanetSpec]
    insts: List[InstSpec]
    has_jitter: bool
    has_drift: bool
    has_linpar: bool
    drift: List[SlotSpec]
    tref: SlotSpec
    tref_from_data: bool
    linpar_names: List[str] = field(default_factory=list)
    linpar: List[SlotSpec] = field(default_factory=list)
    tol: float = 1.0e-4          # rvmodel/__init__.py:466
    itmax: int = 10000           # rvmodel/__init__.py:491
    precision: int = _abi.PREC_FP64   # reduced-precision modes are NOT parity modes (include/rvll.h)

    @property
    def ndim(self):
        return len(self.parnames)

    @property
    def nplanets(self):
        return len(self.planets)

    def to_c(self):
        """Build the ctypes rvll_layout; returns (layout, keepalive)."""
        planets = (_abi.Planet * max(1, len(self.planets)))()
        for i, p in enumerate(self.planets):
            planets[i] = _abi.Planet(p.k_kind, p.p_kind, p.ecc_kind, p.anom_kind, p.k.to_c(), p.p.to_c(),
                                     p.e1.to_c(), p.e2.to_c(), p.anom.to_c(), p.epoch.to_c())
        insts = (_abi.Inst * max(1, len(self.insts)))()
        for i, s in enumerate(self.insts):
            insts[i] = _abi.Inst(s.offset.to_c(), s.jitter.to_c())
        lin = (_abi.Slot * max(1, len(self.linpar)))()
        for i, s in enumerate(self.linpar):
            lin[i] = s.to_c()
        L = _abi.Layout()
        L.struct_size = _abi.C.sizeof(_abi.Layout)
        L.ndim = self.ndim
        L.nplanets = len(self.planets)
        L.ninst = len(self.insts)
        L.has_jitter = int(self.has_jitter)
        L.has_drift = int(self.has_drift)
        L.tref_from_data = int(self.tref_from_data)
        L.nlinpar = len(self.linpar)
        for i in range(4):
            L.drift[i] = self.drift[i].to_c()
        L.tref = self.tref.to_c()
        L.planets = _abi.C.cast(planets, _abi.C.POINTER(_abi.Planet))
        L.insts = _abi.C.cast(insts, _abi.C.POINTER(_abi.Inst))
        L.linpar = _abi.C.cast(lin, _abi.C.POINTER(_abi.Slot))
        L.tol = self.tol
        L.itmax = self.itmax
        L.precision = int(self.precision)
        return L, (planets, insts, lin)


def compile_layout(parnames: Sequence[str], fixedpardict: Dict[str, float], insts: Sequence[str],
                   linpar_names: Optional[Sequence[str]] = None) -> ModelLayout:
    """Resolve the reference's name-driven model structure into slots.

    parnames      free parameter names (any order; sorted here like rvmodel:43)
    fixedpardict  {name: value} of fixed parameters
    insts         instrument names in datadict order (rvmodel:46)
    linpar_names  keys of the model's linpar_dict (rvmodel:131-136), in dict order
    Raises KeyError where the reference's log_likelihood would (a required name is
    neither free nor fixed).
    """
    names = sorted(parnames)
    if len(set(names)) != len(names):
        raise ValueError("duplicate parameter names")
    index = {n: i for i, n in enumerate(names)}
    fixed = dict(fixedpardict)

    def has(name):
        return name in fixed or name in index

    def slot(name):
        if name in fixed:                       # fixed overrides free, rvmodel:178
            return SlotSpec(-1, float(fixed[name]))
        if name in index:
            return SlotSpec(index[name], 0.0)
        raise KeyError(name)

    nplanets = sum(1 for n in names if "k1" in n)             # rvmodel:122-124
    has_drift = any("drift" in n for n in names)              # rvmodel:128-129
    has_linpar = any("linpar" in n for n in names)            # rvmodel:131-136
    has_jitter = any("jitter" in n for n in names)            # rvmodel:138-139

    planets = []
    for n in range(1, nplanets + 1):
        pre = f"planet{n}_"
        if has(pre + "k1"):                                   # rvmodel:412-415
            k_kind, k = _abi.K_K1, slot(pre + "k1")
        else:
            k_kind, k = _abi.K_LOGK1, slot(pre + "logk1")
        if has(pre + "period"):                               # rvmodel:417-420
            p_kind, p = _abi.P_PERIOD, slot(pre + "period")
        else:
            p_kind, p = _abi.P_LOGPERIOD, slot(pre + "logperiod")
        if has(pre + "secos"):                                # rvmodel:425-431
            ecc_kind, e1, e2 = _abi.ECC_SECOS_SESIN, slot(pre + "secos"), slot(pre + "sesin")
        elif has(pre + "ecos"):                               # rvmodel:433-439
            ecc_kind, e1, e2 = _abi.ECC_ECOS_ESIN, slot(pre + "ecos"), slot(pre + "esin")
        else:                                                 # rvmodel:441-447
            try:
                ecc_kind, e1, e2 = _abi.ECC_DIRECT, slot(pre + "ecc"), slot(pre + "omega")
            except KeyError:
                raise KeyError("Something is wrong with the eccentricity parametrisation") from None
        if has(pre + "ml0"):                                  # rvmodel:449-454
            anom_kind, anom = _abi.ANOM_ML0, slot(pre + "ml0")
        else:
            anom_kind, anom = _abi.ANOM_MA0, slot(pre + "ma0")
        epoch = slot(pre + "epoch")                           # rvmodel:456
        planets.append(PlanetSpec(k_kind, p_kind, ecc_kind, anom_kind, k, p, e1, e2, anom, epoch))

    inst_specs = []
    for name in insts:                                        # rvmodel:183-192
        offset = slot(f"{name}_offset")
        jitter = slot(f"{name}_jitter") if has_jitter else SlotSpec(-1, 0.0)
        inst_specs.append(InstSpec(str(name), offset, jitter))
    if not inst_specs:
        raise ValueError("at least one instrument is required")

    zero = SlotSpec(-1, 0.0)
    drift = [zero, zero, zero, zero]
    tref, tref_from_data = zero, True
    if has_drift:                                             # rvmodel:242-260
        for i, key in enumerate(("drift_lin", "drift_quad", "drift_cub", "drift_quar")):
            if has(key):
                drift[i] = slot(key)
        if has("drift_tref"):
            tref, tref_from_data = slot("drift_tref"), False

    lin_names, lin_slots = [], []
    if has_linpar and linpar_names:                           # rvmodel:210-212
        for key in linpar_names:
            lin_names.append(str(key))
            lin_slots.append(slot(f"linpar_{key}"))

    return ModelLayout(parnames=names, planets=planets, insts=inst_specs, has_jitter=has_jitter,
                       has_drift=has_drift, has_linpar=has_linpar, drift=drift, tref=tref,
                       tref_from_data=tref_from_data, linpar_names=lin_names, linpar=lin_slots)
