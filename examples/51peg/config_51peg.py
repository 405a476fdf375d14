# A configuration in the reference's python-module format (see evidence_amd/config.py) for the 51 Peg
# RV time series (256 Hamilton epochs; the data file the reference ships with its example and tests).
import numpy as np
from pathlib import Path

HERE = Path(__file__).resolve().parent

rundict = {
    "target": "51Peg",
    "runid": "example",
    "star_params": {"star_mass": (1.11, 0.02)},
    "save_dir": str(HERE / "chains"),
}

datadict = {
    "hamilton": {
        "datafile": str(HERE.parents[1] / "tests" / "golden" / "51Peg.rv"),
        "instrument": "hamilton",
        "kwargs": {"sep": "\t", "skiprows": (1,)},          # forwarded to pandas.read_csv
    }
}

# [value, jump flag (0 = fixed), [prior name, *prior arguments]]
planet1 = {
    "k1": [0.0, 1, ["Jeffreys", 0.1, 100.0]],
    "period": [0.0, 1, ["UniformFrequency", 1, 100]],
    "ecc": [0.1, 1, ["Beta", 0.867, 3.03]],
    "omega": [0.1, 1, ["Uniform", 0.0, 2 * np.pi]],
    "ma0": [0.1, 1, ["Uniform", 0.0, 2 * np.pi]],
    "epoch": [51050, 0],
}
hamilton = {
    "offset": [0.0, 1, ["Uniform", -10, 10]],
    "jitter": [0.75, 1, ["Uniform", 0.0, 50.0]],
}
input_dict = {"planet1": planet1, "hamilton": hamilton}

configdicts = [rundict, input_dict, datadict]
