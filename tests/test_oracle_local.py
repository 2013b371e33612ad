"""Pins the oracle's usearch_local driver (oracle/ugs_oracle.c: AlignMulti, AlignPos, GetAnchor, KeepAR, EStats,
local accept rules, score-ordered hits) against the compiled, unmodified reference: byte-identical -blast6out
(identities, HSP coordinates, e-values, bit scores) on every case of tests/golden/local_manifest.json."""
import ctypes as C

import numpy as np
import pytest

import golden_util as G
import orc


@pytest.mark.parametrize("name", G.local_case_names())
def test_oracle_local_matches_reference_text(name):
    c, db, qs, b6 = G.load_local(name)
    kw = G.local_params_kw(c)
    p = orc.params(is_nucleo=not c["aa"], **kw)
    assert p.local == 1 and p.id_set == (1 if "id" in c else 0)
    odb = orc.OrcDB(p, db.seqs, db.offs)
    before = orc.lib().orc_local_rescore_diffs()
    hits, nh, pool = odb.search(qs.seqs, qs.offs, nthreads=4)
    got = orc.format_blast6_local(orc.lib(), "orc", p, hits, nh, qs.labels(), db.labels())
    assert got == b6
    assert len(hits) == c["n_hits"]
    assert np.all(hits["flags"] == 1)
    # AlignResult::GetRawScore rescoring the path gives the x-drop score back, and local paths start and end on M
    assert orc.lib().orc_local_rescore_diffs() == before
    assert np.all(hits["cols"] == hits["aln_len"])


def test_local_evalue_formula():
    p = orc.params(True, id=None, local_evalue=1e-6)
    e, b = C.c_double(0), C.c_double(0)
    orc.lib().orc_local_evalue(C.byref(p), 302.0, 344, C.byref(e), C.byref(b))     # first line of the hard_small probe
    assert "%.1f" % b.value == "558.8" and "%.2g" % e.value == "2.1e-157"
