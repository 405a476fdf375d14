"""CPU: the numpy/scipy prior oracle (oracle/priors_oracle.py) against the golden `.ppf` vectors the
reference itself produced — this is what pins it before the GPU tests use it at scale."""
import numpy as np
import pytest

import golden
import prior_cases as pc
from oracle import priors_oracle as po

Q, SETS = golden.prior_sets()


@pytest.mark.parametrize("name,args,vals,raised", SETS, ids=[f"{n}{tuple(a)}" for n, a, _, _ in SETS])
def test_prior_oracle_matches_reference_golden(name, args, vals, raised):
    got = po.ppf(name, args, Q)
    ok = ~raised
    # same scipy routines / same grid as the reference: agreement to rounding everywhere the reference returns
    assert pc.rel_err(got[ok], vals[ok]).max() <= 1e-14, (name, args)
    assert np.all(np.isnan(got[raised]))


def test_sorted_uniform_is_sorted_and_in_range():
    x = np.random.default_rng(0).random((1000, 4))
    t = po.sorted_uniform(x, 2.0, 50.0)
    assert np.all(np.diff(t, axis=1) >= 0) and t.min() >= 2.0 and t.max() <= 50.0
    tl = po.sorted_uniform(x, 2.0, 50.0, log=True)
    assert np.all(np.diff(tl, axis=1) >= 0) and tl.min() >= 2.0 and tl.max() <= 50.0
