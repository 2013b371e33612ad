"""Load golden cases (tests/golden/manifest.json): regenerate the seeded inputs, check their
digests, return inputs + the reference's expected blast6/uc text."""
import hashlib
import importlib.util
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))

_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
_mg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mg)


def case_names(pred=None):
    return [n for n, c in sorted(MANIFEST.items()) if pred is None or pred(c)]


def is_big(c):
    """True when the reference takes the Big ranking path for this case."""
    n = c["db_n"] if c["gen"] == "uniform" else c["n_fam"] * c["fam"]
    return n > c.get("big", 100000)


def load(name):
    c = MANIFEST[name]
    db, qs = _mg.make_inputs(c)
    assert _mg.digest(db) == c["db_sha256"], "generator drift (db) for " + name
    assert _mg.digest(qs) == c["q_sha256"], "generator drift (queries) for " + name
    b6 = open(os.path.join(GOLD, name + ".b6")).read()
    uc = open(os.path.join(GOLD, name + ".uc")).read()
    return c, db, qs, b6, uc


def params_kw(c):
    kw = dict(strand_both=1 if c.get("strand") == "both" else 0)
    if "big" in c:
        kw["big"] = c["big"]
    if "maxaccepts" in c:
        kw["max_accepts"] = c["maxaccepts"]
    if "maxrejects" in c:
        kw["max_rejects"] = c["maxrejects"]
    return kw
