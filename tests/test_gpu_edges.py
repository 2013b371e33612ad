"""Edge cases of the C-ABI on the GPU: empty batches, one-sequence databases, sequences shorter than a word or made of
wildcards only, queries beyond the device envelope (must fail with an error code, never crash), capacity errors."""
import numpy as np
import pytest

import orc
from usearch12_amd import capi, synth

pytestmark = pytest.mark.gpu


def pack(seqs):
    offs = np.zeros(len(seqs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    return np.frombuffer("".join(seqs).encode(), np.uint8).copy(), offs


def both(db, qs, aa=False, ident=0.9, **kw):
    dseq, doff = pack(db)
    qseq, qoff = pack(qs)
    g = capi.UgsDB(capi.params(is_nucleo=not aa, id=ident, **kw), dseq, doff, device=0).search(qseq, qoff)
    o = orc.OrcDB(orc.params(is_nucleo=not aa, id=ident, **kw), dseq, doff).search(qseq, qoff)
    assert np.array_equal(g[1], o[1])
    for f in g[0].dtype.names:
        if f != "cigar_off":
            assert np.array_equal(g[0][f], o[0][f]), f
    return g


def test_empty_query_batch():
    dseq, doff = pack(["ACGTACGTACGTTTGACCA" * 5])
    db = capi.UgsDB(capi.params(True, 0.9), dseq, doff, device=0)
    hits, nh, pool = db.search(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert len(hits) == 0 and len(nh) == 0


def test_single_sequence_database_and_tiny_queries():
    t = "ACGTTGCAAGGCTTACCGATAGGCTATTCGATCGGATATCGCGATATAGCGCTATAGCTAGGATCGATCGGCTAGCTA"
    both([t], [t, t[:30], t[3:60], "ACG", "A", "NNNNNNNNNNNNNNNNNNNN", "acgttgcaaggcttaccgat", t[::-1]], big=0)
    both([t], [t, "ACGTACGT", "NNNNACGTNNNN"])                     # small ranking path


def test_wildcard_only_and_masked_database_sequences():
    db = ["N" * 50, "A" * 60, "ACGTTGCAAGGCTTACCGATAGGCTATTCGATCGGATATCGCGATATAGCG", "AC" * 30]
    both(db, [db[2], "A" * 60, "AC" * 30, "N" * 50], big=0, strand_both=1)


def test_protein_short_sequences():
    t = "MKVLAAGIVGLCAKQWERTYHPLMNDFSCV"
    both([t, t[:4], "X" * 20], [t, t[5:25], "MKV", "XXXXXXXX", t.lower()], aa=True, ident=0.5, big=0)


def test_protein_queries_with_repeats_and_full_word_buckets():
    """aa HSP words (8000) are looked up through a table of 1000 buckets of eight words, <= 15 query positions per bucket: a query
    whose bucket is fuller (a homopolymer run, a short tandem repeat) sorts and searches the general way; a word with more than
    MaxReps = 8 positions keeps its first eight either way (hspfinder.cpp SetA)"""
    rng = np.random.default_rng(41)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    rnd = lambda n: "".join(aa[i] for i in rng.integers(0, 20, n))
    core = [rnd(120) for _ in range(6)]
    t0 = core[0] + "A" * 26 + core[1]                    # 24 x the word AAA: a bucket over its limit
    t1 = core[2] + "ACD" * 7 + core[3]                    # three words seven times each, all in one bucket (ACD, CDA, DAC differ: other buckets) -> repeats within MaxReps
    t2 = core[4] + "LKLKLKLKLKLKLKLKLKLKLKLK" + core[5]    # two words 11 x each (more than MaxReps, fewer than 16)
    t3 = rnd(60) + "AAAC" * 5 + rnd(60)                    # AAA, AAC in the SAME bucket (8 consecutive codes), 5 + 5 positions
    db = [t0, t1, t2, t3] + [rnd(200) for _ in range(20)]
    mut = lambda s, k: "".join((aa[(aa.index(c) + 1) % 20] if i % k == k - 1 else c) for i, c in enumerate(s))
    qs = [t0, t1, t2, t3, mut(t0, 17), mut(t1, 13), mut(t2, 19), mut(t3, 11), t0[100:200], t2[90:], "A" * 40, "LK" * 30]
    both(db, qs, aa=True, ident=0.5, big=0)
    both(db, qs, aa=True, ident=0.8)


def test_long_protein_queries_on_the_small_path():
    """1 500-residue proteins against a small database: the small path samples EVERY word (1 496 rows per query, counts beyond 255), and a
    16-bit counter table of one partition no longer fits the LDS beside the row lists - the plan sizes the tables for 8 bits and the
    kernel walks each partition in sub-ranges (an ad-hoc sweep of the fuzzer's long-sequence mode found the refusal, seed 463); same
    hits as the oracle, both strands of the nt twin as well"""
    for aa, kw in ((True, dict(max_accepts=4, max_rejects=16, bump_pct=10)), (False, dict(strand_both=1, max_accepts=2, max_rejects=8))):
        db, qs = synth.make_hard(77 + int(aa), 8, 10, 30, lmin=1500, lmax=1520, aa=aa)
        g = capi.UgsDB(capi.params(is_nucleo=not aa, id=0.9, **kw), db.seqs, db.offs, device=0).search(qs.seqs, qs.offs)
        o = orc.OrcDB(orc.params(is_nucleo=not aa, id=0.9, **kw), db.seqs, db.offs).search(qs.seqs, qs.offs, nthreads=4)
        assert len(g[0]) > 20 and np.array_equal(g[1], o[1])
        for f in g[0].dtype.names:
            if f != "cigar_off":
                assert np.array_equal(g[0][f], o[0][f]), (aa, f)


def test_query_beyond_envelope_is_an_error_not_a_crash():
    rng = np.random.default_rng(3)
    dseq, doff = pack(["".join("ACGT"[i] for i in rng.integers(0, 4, 300))])
    db = capi.UgsDB(capi.params(True, 0.9), dseq, doff, device=0)
    big_q = "".join("ACGT"[i] for i in rng.integers(0, 4, 70000))
    qseq, qoff = pack([big_q])
    with pytest.raises(capi.UgsError) as e:
        db.search(qseq, qoff)
    assert e.value.code in (-6, -1)
    # the handle stays usable
    qseq, qoff = pack(["".join("ACGT"[i] for i in rng.integers(0, 4, 100))])
    db.search(qseq, qoff)


def test_fetch_capacity_error_reports_demand():
    db0 = synth.make_db(9, 500, 200)
    qs = synth.make_queries(9, db0, 200, 200)
    db = capi.UgsDB(capi.params(True, 0.9), db0.seqs, db0.offs, device=0)
    bat = capi.UgsBatch(db, qs.n, int(qs.offs[-1]))
    bat.upload(qs.seqs, qs.offs); bat.search(); bat.sync()
    import ctypes as C
    hits = np.zeros(qs.n + 1, capi.HIT_DTYPE); nh = np.zeros(qs.n + 1, np.uint32); pool = np.zeros(4, np.uint32)
    used = C.c_uint64(0)
    rc = capi.lib().ugs_batch_fetch(bat.h, hits.ctypes.data, len(hits), nh.ctypes.data, pool.ctypes.data, len(pool), C.byref(used))
    assert rc == -5 and used.value > 4
    h, n, p = bat.fetch()
    assert len(p) == used.value


@pytest.mark.parametrize("kw", [dict(max_accepts=1, max_rejects=0), dict(max_accepts=0, max_rejects=0), dict(max_accepts=40, max_rejects=40),
                                dict(max_accepts=3, max_rejects=256), dict(max_accepts=0, max_rejects=0, strand_both=1, big=0)])
def test_walks_deeper_than_the_kept_candidates_equal_the_oracle(kw):
    """-maxrejects 0 / -maxaccepts 0 (unlimited), 40 + 40, -maxrejects 256 on a database where walks run past the 64 candidates a ranking
    pass keeps (r4: UGS_E_ENVELOPE; r5: the parked walks are continued over the unit's complete sorted list, ugs_deep.hip): every hit
    record equals the oracle's, and the deep stage did run"""
    db, qs = synth.make_hard(77, 12, 100, 160)
    ident = 0.99 if kw.get("max_rejects") == 0 and kw.get("max_accepts") == 1 else 0.93
    p = capi.params(is_nucleo=True, id=ident, **kw)
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
    bat.upload(qs.seqs, qs.offs); bat.search(); bat.sync()
    g = bat.fetch()
    parked, keys = bat.deep_walks()
    assert parked > 0 and keys > 64 * parked
    o = orc.OrcDB(orc.params(is_nucleo=True, id=ident, **kw), db.seqs, db.offs).search(qs.seqs, qs.offs, nthreads=8)
    assert np.array_equal(g[1], o[1])
    for f in g[0].dtype.names:
        if f != "cigar_off":
            assert np.array_equal(g[0][f], o[0][f]), f
    for a, c in zip(g[0][::7], o[0][::7]):
        assert np.array_equal(g[2][int(a["cigar_off"]):int(a["cigar_off"]) + int(a["cigar_len"])], o[2][int(c["cigar_off"]):int(c["cigar_off"]) + int(c["cigar_len"])])
    # the same batch again (the pools have their final size now): the same table
    bat.search(); bat.sync()
    g2 = bat.fetch()
    assert np.array_equal(g[1], g2[1]) and all(np.array_equal(g[0][f], g2[0][f]) for f in g[0].dtype.names if f != "cigar_off")


def test_options_outside_the_envelope_are_refused_at_create():
    db = synth.make_db(5, 200, 120)
    for kw in (dict(local_evalue=1e-3, self=True),               # pair filters exist for usearch_global only
               dict(max_accepts=0, max_rejects=0, termid=0.9, align_flags=4),   # -termid looks at both strands' hits in walk order: not with parked walks
               dict(band=-1)):
        with pytest.raises(capi.UgsError):
            capi.UgsDB(capi.params(is_nucleo=True, id=0.9, **kw), db.seqs, db.offs, device=0)


@pytest.mark.parametrize("limit", [None, "1"])
def test_fully_redundant_database_in_one_partition(limit, monkeypatch):
    """N copies of a sequence in a database of one partition (np = 1 < waves per workgroup): ONE wave of k_rank emits every posting of
    the unit's rows, so its share of the candidate buffer must be able to grow to the whole unit (the regrow was capped
    per workgroup and such a search failed after six identical retries).  UGS_EMIT_LIMIT=1 starts from a one-key buffer."""
    if limit:
        monkeypatch.setenv("UGS_EMIT_LIMIT", limit)
    rng = np.random.default_rng(77)
    rnd = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    t = rnd(400)
    fam = [t] * 700 + [t[:200] + rnd(1) + t[201:] for _ in range(100)]
    db = fam + [rnd(400) for _ in range(50)]
    qs = [t, t[:150] + "A" + t[151:], t[20:380], db[-1]]
    for big in (0, 100):
        g = both(db, qs, ident=0.97, big=big, max_accepts=4, max_rejects=8)
        assert g[1][0] == 4


def test_guard_allocator_and_abort_handler_are_alive():
    """The fault-isolation switches of round 6 (usearch12_amd/csrc/ugs_alloc.cpp) keep working: a child process with UGS_GUARD_ALLOC=2 (every
    device buffer a mapping of its own, right-aligned against an unmapped page), UGS_ABORT_BT and UGS_KERNEL_LOG runs a small search on both
    ranking paths bit-exactly against the oracle, reports guarded allocations, names its kernels - and no kernel steps outside its buffers."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, os, ctypes
sys.path.insert(0, os.path.join(%r, "tests")); sys.path.insert(0, %r)
import numpy as np
from usearch12_amd import capi, synth
import orc
for big in (100, 100000):
    db = synth.make_db(77, 4000, 250); qs = synth.make_queries(77, db, 300, 250)
    kw = dict(is_nucleo=True, id=0.97, big=big)
    h, nh, pool = capi.UgsDB(capi.params(**kw), db.seqs, db.offs, device=0).search(qs.seqs, qs.offs)
    oh, onh, opool = orc.OrcDB(orc.params(**kw), db.seqs, db.offs).search(qs.seqs, qs.offs)
    assert np.array_equal(nh, onh)
    for f in h.dtype.names:
        if f != "cigar_off":
            assert np.array_equal(h[f], oh[f]), f
out = (ctypes.c_ulonglong * 5)()
capi.lib().ugs_debug_alloc_stats(out)
print("ALLOC", list(out))
''' % (ROOT, ROOT)
    env = dict(os.environ, UGS_GUARD_ALLOC="2", UGS_ABORT_BT="stderr", UGS_KERNEL_LOG="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    st = [int(x) for x in r.stdout.split("ALLOC")[1].strip().strip("[]").split(",")]
    assert st[0] == 2 and st[1] > 40 and st[2] > 10 and st[4] > 0, st
    assert "[ugs] kernel k_align launched" in r.stderr and "[ugs] kernel k_align done" in r.stderr and "Memory access fault" not in r.stderr
