"""evidence_amd — MI355X-native RV log-likelihood engine for nested sampling.

A from-scratch replacement for ONE hot path of nicochunger/evidence: the
per-live-point Keplerian RV forward model + Gaussian log-likelihood
(evidence/rvmodel) and the unit-cube prior transform (evidence/priors.py), behind
the reference's own sampler callback signatures.  Host code is plain Python over a
ctypes C-ABI (include/rvll.h) into hand-written HIP kernels for gfx950.  There is
no CPU fallback: without librvll.so and a HIP device the constructors raise.
"""
from ._abi import (FLAG_INVALID_ORBIT, FLAG_NONCONVERGED, FLAG_WANDERED, RvllError, RvllLibraryError)
from .data import EpochTable
from .layout import ModelLayout, compile_layout
from .priors import PriorError, PriorSpec, prior_constructor
from .engine import GpuRVModel

__all__ = ["GpuRVModel", "EpochTable", "ModelLayout", "compile_layout", "PriorSpec", "PriorError",
           "prior_constructor", "RvllError", "RvllLibraryError", "FLAG_INVALID_ORBIT", "FLAG_NONCONVERGED",
           "FLAG_WANDERED"]
__version__ = "0.1.0"
