"""Pins the oracle's pair filters (Accepter::RejectPair: -self -notself -selfid -min_sizeratio -minqt -maxqt -minsl -maxsl)
and -abskew against the compiled, unmodified reference: byte-identical -blast6out on both ranking paths (a rejected pair is
a reject for the terminator on the Big path and invisible to it on the small path)."""
import numpy as np
import pytest

import golden_util as G
import orc


@pytest.mark.parametrize("name", G.pair_case_names())
def test_oracle_pair_filters_match_reference_text(name):
    c, db, qs, b6 = G.load_pairs(name)
    p = orc.params(is_nucleo=True, id=c["id"], **G.pair_params_kw(c))
    assert p.pair_mask or p.filter_mask
    odb = orc.OrcDB(p, db.seqs, db.offs)
    tk, tz, qk, qz = G.pair_keys(db, qs)
    odb.set_pair_keys(tk, tz)
    odb.set_query_pair_keys(qk, qz)
    hits, nh, pool = odb.search(qs.seqs, qs.offs, nthreads=4)
    qlens = np.diff(qs.offs.astype(np.int64))
    got, _ = orc.format_outputs(orc.lib(), "orc", hits, nh, pool, qs.labels(), qlens, db.labels(), True)
    assert got == b6
    assert len(hits) == c["n_hits"]
