"""Config / data ingest compatible with the reference's python-module configs, so that an existing
user configuration runs unmodified on this engine.

Format (evidence/examples/51Peg/config_51Peg_example.py:43-60): a python file defining
`configdicts = [rundict, input_dict, datadict]` where
    input_dict[object][parameter] = [value, jump_flag, [PriorName, *args]]   (flag 0 => fixed)
    datadict[instrument] = {'datafile': path, 'instrument': name, 'kwargs': {... pandas.read_csv ...}}
Semantics follow evidence/config.py:8-148: optional `nplanets` clones a single `planet1` dictionary or
truncates several; returns (rundict, datadict, priordict, fixedpardict) with the data tables loaded
under datadict[inst]['data'].  The priordict holds evidence_amd PriorSpec objects (device transforms).
"""
import importlib.util
from pathlib import Path

from .priors import prior_constructor


def _load_module(configfile):
    path = Path(configfile)
    spec = importlib.util.spec_from_file_location(f"_rvll_config_{abs(hash(str(path.resolve())))}", path)
    if spec is None or spec.loader is None:
        raise ImportError(f"cannot import configuration file {configfile}")
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)          # a fresh module every call: no sys.path edits, no caching
    return module


def read_config(configfile, nplanets=None):
    c = _load_module(configfile)
    rundict, inputdict, datadict = (dict(d) for d in c.configdicts)

    if nplanets is not None:                                           # evidence/config.py:26-58
        if type(nplanets) is not int:
            raise TypeError("nplanets has to be an integer.")
        if nplanets < 0:
            raise ValueError("nplanets has to be positive.")
        planet_keys = [k for k in inputdict if "planet" in k]
        if len(planet_keys) > 1 and nplanets > len(planet_keys):
            raise ValueError("Not enough planet dictionaries for the requested number of planets.")
        if len(planet_keys) == 1:                                      # clone the single planet dictionary
            template = dict(inputdict["planet1"])
            del inputdict["planet1"]
            for n in range(1, nplanets + 1):
                inputdict[f"planet{n}"] = dict(template)
        elif len(planet_keys) > 1:
            for n in range(nplanets + 1, len(planet_keys) + 1):
                del inputdict[f"planet{n}"]
        rundict["nplanets"] = nplanets

    priordict = prior_constructor(inputdict)                           # evidence/config.py:61
    rundict["prior_names"] = {                                         # evidence/config.py:117-148
        f"{obj}_{par}": f"{entry[2][0]}: {entry[2][1:]}"
        for obj, pars in inputdict.items() for par, entry in pars.items()
        if isinstance(entry, list) and entry[1] != 0}
    datadict = {inst: dict(spec) for inst, spec in datadict.items()}
    read_data(datadict)                                                # evidence/config.py:67
    return rundict, datadict, priordict, get_fixedparvalues(inputdict)


def get_parnames(inputdict):
    """(free names, fixed names), evidence/config.py:75-87."""
    free, fixed = [], []
    for obj, pars in inputdict.items():
        for par, entry in pars.items():
            (free if entry[1] > 0 else fixed if entry[1] == 0 else []).append(f"{obj}_{par}")
    return free, fixed


def get_fixedparvalues(inputdict):
    """{name: value} of flag-0 parameters, evidence/config.py:90-99."""
    return {f"{obj}_{par}": entry[0] for obj, pars in inputdict.items() for par, entry in pars.items()
            if entry[1] == 0}


def read_data(datadict):
    """Load every instrument's data file with pandas.read_csv(**kwargs), evidence/config.py:102-114."""
    import pandas as pd
    for inst, spec in datadict.items():
        spec["data"] = pd.read_csv(spec["datafile"], **spec.get("kwargs", {}))
