"""Pins the oracle (oracle/ugs_oracle.c) against golden vectors produced by the compiled,
unmodified reference: byte-identical -blast6out and -uc text on every case."""
import numpy as np
import pytest

import golden_util as G
import orc


@pytest.mark.parametrize("name", G.case_names())
def test_oracle_matches_reference_text(name):
    c, db, qs, b6, uc = G.load(name)
    p = orc.params(is_nucleo=not c["aa"], id=c["id"], **G.params_kw(c))
    odb = orc.OrcDB(p, db.seqs, db.offs)
    hits, nh, pool = odb.search(qs.seqs, qs.offs, nthreads=4)
    qlens = np.diff(qs.offs.astype(np.int64))
    ob6, ouc = orc.format_outputs(orc.lib(), "orc", hits, nh, pool, qs.labels(), qlens, db.labels(), not c["aa"])
    assert ob6 == b6
    assert ouc == uc
    assert len(hits) == c["n_hits"]


def test_oracle_threads_equal_single():
    c, db, qs, b6, uc = G.load("hard_big")
    p = orc.params(is_nucleo=True, id=c["id"], **G.params_kw(c))
    odb = orc.OrcDB(p, db.seqs, db.offs)
    h1, n1, p1 = odb.search(qs.seqs, qs.offs, nthreads=1)
    h4, n4, p4 = odb.search(qs.seqs, qs.offs, nthreads=3)
    assert np.array_equal(n1, n4)
    assert h1.tobytes() == h4.tobytes()
    assert np.array_equal(p1, p4)
