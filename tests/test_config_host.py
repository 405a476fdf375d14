"""CPU: config/data ingest against the reference's own tests of its reader (tests/test_config.py:7-76),
on a config in the same python-module format."""
from pathlib import Path

import numpy as np
import pytest

from evidence_amd import config
from evidence_amd.data import EpochTable
from evidence_amd.layout import compile_layout

CFG = Path(__file__).resolve().parents[1] / "examples" / "51peg" / "config_51peg.py"


def test_nplanets_type():                                  # tests/test_config.py:7-19
    with pytest.raises(TypeError):
        config.read_config(CFG, nplanets=0.5)
    with pytest.raises(ValueError):
        config.read_config(CFG, nplanets=-6)


def test_no_planets_argument():                            # tests/test_config.py:21-37
    rundict, datadict, priordict, fixeddict = config.read_config(CFG)
    assert len(priordict) == 7 and len(fixeddict) == 1
    assert len(datadict["hamilton"]["data"]) == 256
    assert rundict["target"] == "51Peg" and rundict["star_params"]["star_mass"] == (1.11, 0.02)
    assert "nplanets" not in rundict
    assert rundict["prior_names"]["planet1_ecc"] == "Beta: [0.867, 3.03]"


def test_with_planets():                                   # tests/test_config.py:39-76
    for n in (0, 1, 2):
        rundict, datadict, priordict, fixeddict = config.read_config(CFG, nplanets=n)
        assert rundict["nplanets"] == n
        assert sum("k1" in p for p in priordict) == n
        for k in range(1, n + 1):
            assert f"planet{k}_period" in priordict and f"planet{k}_epoch" in fixeddict


def test_repeated_reads_do_not_share_state():
    a = config.read_config(CFG, nplanets=2)
    b = config.read_config(CFG)                            # the reference's importlib caching would leak planet2 here
    assert "planet2_k1" in a[2] and "planet2_k1" not in b[2]


def test_config_feeds_the_layout_compiler():
    rundict, datadict, priordict, fixeddict = config.read_config(CFG, nplanets=1)
    table = EpochTable.from_datadict(datadict)
    assert table.n_epochs == 256 and table.insts == ["hamilton"]
    layout = compile_layout(list(priordict), fixeddict, table.insts)
    assert layout.parnames == ["hamilton_jitter", "hamilton_offset", "planet1_ecc", "planet1_k1", "planet1_ma0",
                               "planet1_omega", "planet1_period"]
    free, fixed = config.get_parnames({"planet1": {"k1": [0.0, 1, ["Uniform", 0, 1]], "epoch": [5, 0]}})
    assert free == ["planet1_k1"] and fixed == ["planet1_epoch"]
    assert np.isclose(table.time[0], 50002.665695)
