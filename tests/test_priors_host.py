"""CPU: prior specifications (names, arguments, tables) against the golden vectors produced by the
reference's own `.ppf` calls, and the reference's two prior unit tests (tests/test_priors.py:13-15)."""
import numpy as np
import pytest

import golden
import prior_cases as pc
from evidence_amd import _abi
from evidence_amd import priors as P

Q, SETS = golden.prior_sets()
TABLE_SETS = [(n, a, v, r) for (n, a, v, r) in SETS if pc.spec_for(n, a).kind == _abi.PRIOR_TABLE]


def test_every_reference_prior_name_is_known():
    # evidence/priors.py:429-467
    for name in ["Uniform", "Jeffreys", "ModJeffreys", "UniformFrequency", "Normal", "LogNormal", "Log10Normal",
                 "Binormal", "AsymmetricNormal", "TruncatedUNormal", "TruncatedRayleigh", "PowerLaw",
                 "DoublePowerLaw", "Sine", "Alpha", "Beta", "Gamma", "SortedUniform", "SortedLogUniform"]:
        assert name in P.distdict


def test_prior_constructor_walk_matches_reference_semantics():
    # evidence/priors.py:472-505 on the shipped 51Peg config (config_51Peg_example.py:43-56)
    input_dict = {"planet1": {"k1": [0.0, 1, ["Jeffreys", 0.1, 100.0]], "period": [0.0, 1, ["UniformFrequency", 1, 100]],
                              "ecc": [0.1, 1, ["Beta", 0.867, 3.03]], "omega": [0.1, 1, ["Uniform", 0.0, 2 * np.pi]],
                              "ma0": [0.1, 1, ["Uniform", 0.0, 2 * np.pi]], "epoch": [51050, 0]},
                  "hamilton": {"offset": [0.0, 1, ["Uniform", -10, 10]], "jitter": [0.75, 1, ["Uniform", 0.0, 50.0]]},
                  "notalist": {"x": 3.0}}
    pd = P.prior_constructor(input_dict)
    assert sorted(pd) == ["hamilton_jitter", "hamilton_offset", "planet1_ecc", "planet1_k1", "planet1_ma0",
                          "planet1_omega", "planet1_period"]            # 7 free, epoch (flag 0) skipped
    assert pd["planet1_ecc"].kind == _abi.PRIOR_BETA and pd["planet1_k1"].args[:2] == (0.1, 100.0)
    with pytest.raises(P.PriorError):
        P.prior_constructor({"a": {"b": [0.0, 1, ["NoSuchPrior", 1, 2]]}})


def test_invalid_arguments_raise():
    for bad in (lambda: P.Uniform(2, 1), lambda: P.Jeffreys(0, 1), lambda: P.Beta(-1, 2), lambda: P.Normal(0, 0)):
        with pytest.raises(P.PriorError):
            bad()


@pytest.mark.parametrize("name,args,vals,raised", TABLE_SETS, ids=[f"{n}{tuple(a)}" for n, a, _, _ in TABLE_SETS])
def test_table_knots_reproduce_reference_interpolation(name, args, vals, raised):
    """The reference inverts its CDF grid with scipy's interp1d, which evaluates 1-D linear tables through
    numpy.interp; applying numpy.interp to OUR knots must give the golden values wherever the reference
    returned a value, and our knot range must end where the reference raised."""
    spec = pc.spec_for(name, args)
    cdf, x = spec.table_cdf, spec.table_x
    assert np.all(np.diff(cdf) >= 0)
    wrapped = spec.args[2] != 0
    inside = (Q >= cdf[0]) & (Q <= cdf[-1])
    if wrapped:
        inside &= (Q > 0) & (Q < 1)
    got = np.interp(Q[inside], cdf, x)
    ok = ~raised[inside]
    assert pc.rel_err(got[ok], vals[inside][ok]).max() <= 1e-13
    # where the reference raised ValueError (q outside the grid's cdf range) we are outside too
    interior = (Q > 0) & (Q < 1) if wrapped else np.ones_like(Q, bool)
    assert not np.any(raised & inside & interior)
    assert np.all(raised[interior & ~inside])


def test_reference_unit_test_values():
    # tests/test_priors.py:13-15: Uniform(4,6).ppf(0.5|0|1) = 5|4|6 ; golden set 0 is Uniform(4, 6)
    name, args, vals, _ = SETS[0]
    assert (name, args) == ("Uniform", [4.0, 6.0])
    for q, want in ((0.5, 5.0), (0.0, 4.0), (1.0, 6.0)):
        assert vals[np.flatnonzero(Q == q)[0]] == want
