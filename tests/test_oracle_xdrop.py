"""Pins the oracle's gapped x-drop restatement (oracle/ugs_oracle.c xd_*) to the reference:
tests/golden/xdrop_{nt,aa}.txt hold answers of the reference's own XDropFwdFastMem /
XDropBwdFastMem / XDropAlignMem (see tests/golden/make_golden_xdrop.py), plus the reference's
built-in known-answer test (cmd_test, xdropalignmem.cpp:338-364: SEQVENCE/SEQVECE -> 27.0,
Leni 8, Lenj 7, MMMMMDMM)."""
import pytest

import golden_util
import orc
from usearch12_amd.abi import XDROP_ALIGN, XDROP_FWD, XDROP_BWD

MODE = {"A": XDROP_ALIGN, "F": XDROP_FWD, "B": XDROP_BWD}


def test_reference_kat():
    p = orc.xdrop_params(False, 32.0)
    sc, loi, loj, li, lj, path, _ = orc.xdrop_job(p, "SEQVENCE", "SEQVECE", XDROP_FWD)
    assert (sc, li, lj, path) == (27.0, 8, 7, "MMMMMDMM")


@pytest.mark.parametrize("name", ["nt", "aa"])
def test_oracle_matches_reference(name):
    cases = golden_util.load_xdrop(name)
    assert len(cases) > 600
    bad = []
    for k, c in enumerate(cases):
        p = orc.xdrop_params(name == "nt", c["x"])
        got = orc.xdrop_job(p, c["a"], c["b"], MODE[c["mode"]], c["anc"])
        assert got is not None
        sc, loi, loj, li, lj, path, _ = got
        w = c["want"]
        if c["mode"] == "A":
            ok = (sc, loi, loj, li, lj, path) == w
        else:
            ok = (sc, li, lj, path) == (w[0], w[3], w[4], w[5])
        if not ok:
            bad.append((k, c["mode"], got[:5], w[:5]))
    assert not bad, bad[:5]
