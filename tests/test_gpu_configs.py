"""The BASELINE.json configurations at their literal database sizes on one GPU (C1: 10k x 250 nt vs 50k, small ranking path;
one C4 shard: 125k of 10M x 250 nt queries vs the full 5M-sequence DB, i.e. the share of one of 8 GPUs at 1/10 of its length;
C5: 1M x 300 aa vs 2M aa, -id 0.8): the whole query set runs on the GPU, a sample of it through the oracle (its
single-threaded index build over the full database dominates the run time), every hit record and sampled path compared;
plus size-independent properties over the full hit table."""
import os

import numpy as np
import pytest

import orc
from usearch12_amd import capi, synth

pytestmark = pytest.mark.gpu


def _check(db, qs, aa, ident, n_oracle, min_hit_frac):
    p = capi.params(is_nucleo=not aa, id=ident)
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
    bat.upload(qs.seqs, qs.offs)
    bat.search(); bat.sync()
    hits, nh, pool = bat.fetch()
    # properties over the whole table: one accept per query, identities at or above the threshold, sources recovered
    assert nh.max() <= 1 and len(hits) == int(nh.sum())
    assert len(hits) >= min_hit_frac * qs.n
    ident_f = hits["ids"].astype(np.float64) / np.maximum(hits["aln_len"], 1)
    assert ident_f.min() >= float(np.float32(ident)) - 1e-12
    assert np.all(hits["target"] < db.n) and np.all(np.diff(hits["query"].astype(np.int64)) > 0)
    src = qs.src[hits["query"]]
    assert np.mean(src == hits["target"]) > 0.99          # a mutated copy finds the sequence it was made from
    # the oracle on a prefix of the queries
    odb = orc.OrcDB(orc.params(is_nucleo=not aa, id=ident), db.seqs, db.offs)
    so = qs.offs[:n_oracle + 1].copy()
    oh, onh, opool = odb.search(qs.seqs[:int(so[-1])], so, nthreads=min(64, os.cpu_count() or 1))
    k = int(nh[:n_oracle].sum())
    assert np.array_equal(nh[:n_oracle], onh)
    for f in hits.dtype.names:
        if f != "cigar_off":
            assert np.array_equal(hits[:k][f], oh[f]), f
    for a, b in zip(hits[:k][::17], oh[::17]):
        assert np.array_equal(pool[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["cigar_len"])],
                              opool[int(b["cigar_off"]):int(b["cigar_off"]) + int(b["cigar_len"])])


def test_c1_literal_size_small_ranking_path():
    db = synth.make_db(1, 50_000, 250)
    qs = synth.make_queries(1, db, 10_000, 250)
    _check(db, qs, False, 0.97, 2_000, 0.75)


def test_c4_shard_against_the_full_5m_database():
    db = synth.make_db(4, 5_000_000, 250)
    qs = synth.make_queries(4, db, 125_000, 250)
    _check(db, qs, False, 0.97, 1_500, 0.75)


def test_c5_protein_literal_size():
    db = synth.make_db(5, 2_000_000, 300, aa=True)
    qs = synth.make_queries(5, db, 1_000_000, 300, aa=True)
    _check(db, qs, True, 0.8, 1_500, 0.75)
