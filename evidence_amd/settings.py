"""Sampler settings as configuration only: the defaults and the type checks of the reference's two
wrappers, so that a run configured for the reference behaves the same here.  No sampler orchestration, no output
directories, no pickling — those stay with whoever drives the sampler (out of scope).

    polychord_defaults(ndim, polysettings)   evidence/polychord/__init__.py:330-373 (set_polysettings)
    ultranest_defaults(ndim, ultrasettings)  evidence/ultranest/__init__.py:333-345 (set_ultrasettings)
"""
from typing import Optional

# setting -> required type, exactly the five the reference checks (evidence/polychord/__init__.py:344-371); its
# messages say "integer" / "boolean" / "float"
_POLY_TYPES = (("nlive", int, "an integer"), ("num_repeats", int, "an integer"), ("do_clustering", bool, "a boolean"),
               ("read_resume", bool, "a boolean"), ("precision_criterion", float, "a float"))


def polychord_defaults(ndim: int, polysettings: Optional[dict] = None) -> dict:
    """The PolyChord settings of a run: nlive = 25 ndim, num_repeats = 5 ndim, clustering on, no resume files,
    feedback 1, precision_criterion 0.001, boost_posterior 0 — updated with the user's dictionary, whose entries are
    type-checked as the reference does (`type(x) is not T` -> TypeError, so a bool is not an int here either)."""
    settings = {"nlive": 25 * ndim, "num_repeats": 5 * ndim, "do_clustering": True, "write_resume": False,
                "read_resume": False, "feedback": 1, "precision_criterion": 0.001, "boost_posterior": 0.0}
    if polysettings is not None:
        if type(polysettings) is not dict:
            raise TypeError("polysettings has to be a dictionary")
        for name, typ, word in _POLY_TYPES:
            if name in polysettings and type(polysettings[name]) is not typ:
                raise TypeError(f"{name} has to be {word} (got type {type(polysettings[name])})")
        settings.update(polysettings)
    return settings


def ultranest_defaults(ndim: int, ultrasettings: Optional[dict] = None) -> dict:
    """The UltraNest settings of a run: nlive = 25 ndim, nsteps = 3 ndim (slice-sampler moves per new point),
    dlogz 0.5, frac_remain 0.01, num_bootstraps 30 — updated with the user's dictionary (the reference checks only
    that it IS a dictionary)."""
    settings = {"nlive": 25 * ndim, "nsteps": 3 * ndim, "dlogz": 0.5, "frac_remain": 0.01, "num_bootstraps": 30}
    if ultrasettings is not None:
        if type(ultrasettings) is not dict:
            raise TypeError("ultrasettings has to be a dictionary")
        settings.update(ultrasettings)
    return settings
