"""Pair filters (Accepter::RejectPair) and -abskew on the GPU (-m gpu): k_align<true> through the C-ABI vs the reference's
golden -blast6out and the oracle's records, on both ranking paths; the CLI; loud failure when a small-path walk wants more
candidates than the device keeps."""
import os
import subprocess

import numpy as np
import pytest

import golden_util as G
import orc
from usearch12_amd import capi

pytestmark = pytest.mark.gpu


def _run(c, db, qs, **extra):
    kw = G.pair_params_kw(c)
    kw.update(extra)
    p = capi.params(is_nucleo=True, id=c["id"], **kw)
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    tk, tz, qk, qz = G.pair_keys(db, qs)
    gdb.set_pair_keys(tk, tz)
    return p, gdb.search(qs.seqs, qs.offs, pair_keys=(qk, qz))


@pytest.mark.parametrize("name", G.pair_case_names())
def test_gpu_pair_filters_match_reference_text(name):
    c, db, qs, b6 = G.load_pairs(name)
    p, (hits, nh, pool) = _run(c, db, qs)
    qlens = np.diff(qs.offs.astype(np.int64))
    got, _ = orc.format_outputs(capi.lib(), "ugs", hits, nh, pool, qs.labels(), qlens, db.labels(), True)
    assert got == b6


@pytest.mark.parametrize("name", ["pair_selfid_s", "pair_size_b", "pair_len_s"])
def test_gpu_pair_filters_equal_oracle_records(name):
    c, db, qs, b6 = G.load_pairs(name)
    p, (hits, nh, pool) = _run(c, db, qs)
    odb = orc.OrcDB(orc.params(is_nucleo=True, id=c["id"], **G.pair_params_kw(c)), db.seqs, db.offs)
    tk, tz, qk, qz = G.pair_keys(db, qs)
    odb.set_pair_keys(tk, tz); odb.set_query_pair_keys(qk, qz)
    oh, onh, opool = odb.search(qs.seqs, qs.offs, nthreads=4)
    assert np.array_equal(nh, onh)
    for f in hits.dtype.names:
        if f != "cigar_off":
            assert np.array_equal(hits[f], oh[f]), f


def test_pair_filters_need_their_keys():
    c, db, qs, b6 = G.load_pairs("pair_self_b")
    p = capi.params(is_nucleo=True, id=c["id"], **G.pair_params_kw(c))
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    with pytest.raises(capi.UgsError) as e:
        gdb.search(qs.seqs, qs.offs)
    assert e.value.code == -1 and "pair" in str(e.value)
    c, db, qs, b6 = G.load_pairs("pair_size_b")
    p = capi.params(is_nucleo=True, id=c["id"], **G.pair_params_kw(c))
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    tk, tz, qk, qz = G.pair_keys(db, qs)
    tz = tz.copy(); tz[3] = 0xffffffff
    with pytest.raises(capi.UgsError) as e:
        gdb.set_pair_keys(tk, tz)
    assert "size=" in str(e.value)


def test_small_path_walk_deeper_than_kept_candidates_is_loud():
    """-notself on the small path passes over every other target without counting it: the reference walks its whole list,
    the device keeps 64 candidates and must say so instead of returning a shortened walk"""
    c, db, qs, b6 = G.load_pairs("pair_notself_b")
    kw = G.pair_params_kw(c)
    kw.pop("big")
    p = capi.params(is_nucleo=True, id=c["id"], **kw)
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    tk, tz, qk, qz = G.pair_keys(db, qs)
    gdb.set_pair_keys(tk, tz)
    try:
        hits, nh, pool = gdb.search(qs.seqs, qs.offs, pair_keys=(qk, qz))
    except capi.UgsError as e:
        assert e.code == -6 and "candidates" in str(e)
        return
    # no walk needed more than 64 candidates: then the result must be the oracle's
    odb = orc.OrcDB(orc.params(is_nucleo=True, id=c["id"], **kw), db.seqs, db.offs)
    odb.set_pair_keys(tk, tz); odb.set_query_pair_keys(qk, qz)
    oh, onh, _ = odb.search(qs.seqs, qs.offs, nthreads=4)
    assert np.array_equal(nh, onh) and np.array_equal(hits["target"], oh["target"])


def test_small_path_selfid_among_many_identical_sequences_walks_on():
    """-selfid on the small path passes over a target that equals the query without counting it (searcher.cpp:63-67): among 150 copies
    of the query the walk needs candidate 151 and beyond.  r4 kept 32 spare candidates and failed loudly (UGS_ERR_PAIRCAP) past them; r5
    parks such a walk and continues it over the complete list (deep walks): the oracle's records."""
    rng = np.random.default_rng(91)
    rnd = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    base = rnd(300)
    near = [base[:k] + ("A" if base[k] != "A" else "C") + base[k + 1:] for k in (40, 120, 200, 260)]
    seqs = [base] * 150 + near + [rnd(300) for _ in range(200)]
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]
    offs = np.zeros(len(seqs) + 1, np.uint64); offs[1:] = np.cumsum([len(x) for x in seqs])
    dseq = np.frombuffer("".join(seqs).encode(), np.uint8).copy()
    q = [base, near[0], rnd(300)]
    qoff = np.zeros(len(q) + 1, np.uint64); qoff[1:] = np.cumsum([len(x) for x in q])
    qseq = np.frombuffer("".join(q).encode(), np.uint8).copy()
    kw = dict(selfid=True, max_accepts=2, max_rejects=8)
    p = capi.params(is_nucleo=True, id=0.9, **kw)
    gdb = capi.UgsDB(p, dseq, offs, device=0)
    bat = capi.UgsBatch(gdb, len(q), int(qoff[-1]))
    bat.upload(qseq, qoff); bat.search(); bat.sync()
    hits, nh, pool = bat.fetch()
    assert bat.deep_walks()[0] >= 1
    oh, onh, _ = orc.OrcDB(orc.params(is_nucleo=True, id=0.9, **kw), dseq, offs).search(qseq, qoff)
    assert np.array_equal(nh, onh) and nh[0] == 2
    for f in hits.dtype.names:
        if f != "cigar_off":
            assert np.array_equal(hits[f], oh[f]), f


def test_cli_pair_filters_identical_to_reference(tmp_path):
    cli = os.path.join(os.path.dirname(capi.LIB_PATH), "ugs_cli")
    for name in ("pair_self_s", "pair_size_b", "pair_len_b", "pair_selfid_b"):
        c, db, qs, b6 = G.load_pairs(name)
        dbfa, qfa, out = str(tmp_path / "db.fa"), str(tmp_path / "q.fa"), str(tmp_path / "o.b6")
        db.write_fasta(dbfa); qs.write_fasta(qfa)
        cmd = [cli, "-usearch_global", qfa, "-db", dbfa, "-id", str(c["id"]), "-strand", c["strand"], "-blast6out", out, "-batch", "300"]
        for opt in ("big", "maxaccepts", "maxrejects"):
            if opt in c:
                cmd += ["-" + opt, str(c[opt])]
        for k, v in c["opts"].items():
            cmd += ["-" + k] + ([] if v is None else [str(v)])
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        assert open(out).read() == b6, name
