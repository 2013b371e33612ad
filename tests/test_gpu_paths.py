"""Which ranking code a search ran is a tested property: the bitmap kernel (ugs_rank2.hip) must take the
dense Big-path shapes, its deferral path (units handed on to k_rank) must give the same candidates, and searching the same batch
repeatedly must give byte-identical candidate lists and hit tables on every ranking path."""
import os
import zlib

import numpy as np
import pytest

import golden_util as G
import orc
from usearch12_amd import capi, synth

pytestmark = pytest.mark.gpu


def _search(db, qs, env=None, reps=1, **kw):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k); os.environ[k] = v
    try:
        gdb = capi.UgsDB(capi.params(**kw), db.seqs, db.offs, device=0)          # (the debug switches are read at ugs_db_create)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
    bat.upload(qs.seqs, qs.offs)
    outs = []
    for _ in range(reps):
        bat.search(); bat.sync()
        cand, cnt, n = bat.candidates()
        h, nh, pool = bat.fetch()
        K = cand.shape[1]
        mask = np.arange(K)[None, :] < n[:, None]
        crc = zlib.crc32(np.where(mask, cand, 0).tobytes()) ^ zlib.crc32(np.where(mask, cnt, 0).tobytes()) ^ zlib.crc32(n.tobytes())
        for f in ("query", "target", "ids", "mism", "aln_len", "qlo", "qhi", "tlo", "thi", "strand"):
            if f in h.dtype.names:
                crc ^= zlib.crc32(np.ascontiguousarray(h[f]).tobytes())
        outs.append((crc, bat.kernel_hits(), (np.where(mask, cand, 0), np.where(mask, cnt, 0), n.copy())))
    return outs


@pytest.fixture(scope="module")
def c2_small():
    db = synth.make_db(12, 300000, 250)
    qs = synth.make_queries(12, db, 20000, 250)
    return db, qs


def test_bitmap_kernel_takes_the_dense_big_path_and_equals_k_rank(c2_small):
    db, qs = c2_small
    a = _search(db, qs, {"UGS_RANK2": "0"}, is_nucleo=True, id=0.97)[0]
    b = _search(db, qs, None, is_nucleo=True, id=0.97)[0]
    assert a[1]["r2_launched"] == 0 and a[1]["r2_units"] == 0
    assert b[1]["r2_launched"] == 1 and b[1]["r2_units"] > 0.95 * qs.n and b[1]["r2_units"] + b[1]["deferred"] == qs.n
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(x, y)
    assert a[0] == b[0]


@pytest.mark.parametrize("g", [None, "8192", "24576", "65536"])
def test_bitmap_kernel_over_16_bit_postings_equals_the_32_bit_stream(c2_small, g):
    """r6: a plain search streams the postings as 16-bit offsets inside their partition (the default); UGS_R2_P16=0 keeps the 32-bit
    stream.  Same candidates, counts and hit tables as each other and as k_rank, for the default partition size and for 8 192-target
    (37 partitions, sub-rows of ~30 postings: most chunks start 1-3 elements early), 24 576- and 65 536-target partitions; both strands."""
    db, qs = c2_small
    sub = qs.slice(0, 6000)
    env = {"UGS_RANK2": "1"}
    if g:
        env["UGS_R2_G"] = g
    for kw in (dict(is_nucleo=True, id=0.97), dict(is_nucleo=True, id=0.97, strand_both=1)):
        a = _search(db, sub, {"UGS_RANK2": "0"}, **kw)[0]
        b = _search(db, sub, dict(env, UGS_R2_P16="0"), **kw)[0]
        c = _search(db, sub, env, **kw)[0]
        assert b[1]["r2_kernel"] == "k_rank2" and c[1]["r2_kernel"] == "k_rank2<P16>"
        assert b[1]["r2_units"] == c[1]["r2_units"] and b[1]["deferred"] == c[1]["deferred"] and c[1]["r2_units"] > 0.9 * sub.n * (2 if "strand_both" in kw else 1)
        for x, y, z in zip(a[2], b[2], c[2]):
            assert np.array_equal(x, y) and np.array_equal(x, z)
        assert a[0] == b[0] == c[0]


def test_units_deferred_to_k_rank_give_the_same_candidates(c2_small):
    """a kept-key list of 8 entries defers nearly every unit: the general kernel behind the bitmap kernel then ranks them"""
    db, qs = c2_small
    a = _search(db, qs, None, is_nucleo=True, id=0.97)[0]
    b = _search(db, qs, {"UGS_R2_KCAP": "8"}, is_nucleo=True, id=0.97)[0]
    assert b[1]["r2_launched"] == 1 and b[1]["deferred"] > 0.5 * qs.n and b[1]["r2_units"] + b[1]["deferred"] == qs.n
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(x, y)
    assert a[0] == b[0]


def test_bitmap_kernel_with_small_partitions_and_several_windows(c2_small):
    """8192-target partitions: 37 partitions in several windows of the chunk list, sub-rows of ~30 postings"""
    db, qs = c2_small
    a = _search(db, qs, {"UGS_RANK2": "0"}, is_nucleo=True, id=0.97)[0]
    b = _search(db, qs, {"UGS_R2_G": "8192", "UGS_RANK2": "1"}, is_nucleo=True, id=0.97)[0]
    assert b[1]["r2_units"] > 0.9 * qs.n
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("g", ["8192", "16384", None])
def test_bitmap_kernel_with_a_nearly_full_kept_key_list(c2_small, g):
    """families of 120-200 targets that share a window with the query, spread over the partitions: every member keeps a key (count >= 3 keys are never
    pruned), so the kept-key list - the LAST region of the wave's LDS carve - fills to just under its capacity of 252.  ADVICE r04:
    the host sized the carve with the gather kernel's smaller hash filters, the top of the list lay outside the allocation
    (dropped writes, reads of 0 = a key of count 15 / row 0 / target 0) wherever the 1280-byte LDS granule did not cover it."""
    base, _ = c2_small
    rng = np.random.default_rng(31)
    L = 250
    seqs = base.seqs.reshape(base.n, L).copy()
    q = []
    for f, size in enumerate((120, 140, 160, 180, 200)):      # (+ ~100 unpruned count-2 keys of unrelated targets: 220 ... 300 kept keys against a list of 252)
        # a member shares ONE 75-letter window with its family's prototype (3-4 of the query's ~11 sampled words: a count >= 3 key, which is
        # never pruned, for two or three records - a partition's record list stays far from its 128 entries), the rest of it is unrelated
        proto = seqs[1000 + f].copy()
        idx = rng.choice(base.n - 5000, size=size, replace=False) + 2000
        for t in idx:
            a0 = int(rng.integers(0, L - 75))
            seqs[t, a0:a0 + 75] = proto[a0:a0 + 75]
        for k in range(12):
            row = proto.copy()
            if k:
                pos = rng.integers(0, L, size=1)
                row[pos] = seqs[(1000 + f + 7 * k) % base.n][pos]
            q.append(row)
    db = synth.SeqSet(seqs.reshape(-1), base.offs, lambda i: "t%d" % i)
    qs0 = synth.make_queries(31, db, 500, L)
    qseqs = np.concatenate([qs0.seqs] + q)
    qoffs = np.concatenate([qs0.offs, qs0.offs[-1] + np.arange(1, len(q) + 1, dtype=np.uint64) * np.uint64(L)])
    qs = synth.SeqSet(qseqs, qoffs, lambda i: "q%d" % i)
    env = {"UGS_RANK2": "1", "UGS_LONGROWS": "0"}       # (the families make a few index rows long: keep the dense-index kernels)
    if g:
        env["UGS_R2_G"] = g
    a = _search(db, qs, {"UGS_RANK2": "0", "UGS_LONGROWS": "0"}, is_nucleo=True, id=0.97)[0]
    b = _search(db, qs, env, is_nucleo=True, id=0.97)[0]
    assert b[1]["r2_launched"] == 1 and b[1]["r2_units"] + b[1]["deferred"] == qs.n
    if g:                                               # (small partitions: the families' records fit; the largest families overflow the key list and defer)
        assert b[1]["r2_units"] >= 500 + 24 and b[1]["deferred"] >= 1, b[1]
    n_fam = a[2][2][500:]
    assert n_fam.min() >= 30, n_fam                     # the family queries do see their hundreds of count >= 3 targets
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(x, y)
    assert a[0] == b[0]


def test_bitmap_kernel_both_strands_and_oracle(c2_small):
    db, _ = c2_small
    qs = synth.make_queries(13, db, 600, 250)
    out = _search(db, qs, None, is_nucleo=True, id=0.97, strand_both=1)[0]
    assert out[1]["r2_launched"] == 1
    op = orc.params(is_nucleo=True, id=0.97, strand_both=1)
    odb = orc.OrcDB(op, db.seqs, db.offs)
    cand, cnt, n = out[2]
    for qi in range(0, qs.n, 7):
        q = qs.seqs[int(qs.offs[qi]):int(qs.offs[qi + 1])]
        for s in range(2):
            on, oc, occ = odb.rank(q if s == 0 else orc.revcomp(q), cap=cand.shape[1])
            u = qi * 2 + s
            m = min(on, cand.shape[1])
            assert n[u] == m and np.array_equal(cand[u, :m], oc[:m]) and np.array_equal(cnt[u, :m], occ[:m]), (qi, s)


@pytest.mark.parametrize("shape", ["c2", "c2_krank", "id90", "both", "small", "aa"])
def test_same_batch_ten_times_gives_the_same_bytes(shape, c2_small):
    """determinism of every ranking path: candidate lists and hit tables of ten searches of one uploaded batch"""
    if shape in ("c2", "c2_krank", "id90", "both"):
        db, qs = c2_small
        kw = dict(is_nucleo=True, id=0.9 if shape == "id90" else 0.97, strand_both=1 if shape == "both" else 0)
        env = {"UGS_RANK2": "0"} if shape == "c2_krank" else None
        if shape in ("id90", "both"):
            qs = synth.make_queries(14, db, 4000, 250)
    elif shape == "small":
        db = synth.make_db(15, 50000, 250); qs = synth.make_queries(15, db, 4000, 250); kw = dict(is_nucleo=True, id=0.97); env = None
    else:
        db = synth.make_db(16, 150000, 300, aa=True); qs = synth.make_queries(16, db, 4000, 300, aa=True); kw = dict(is_nucleo=False, id=0.8); env = None
    outs = _search(db, qs, env, reps=10, **kw)
    assert len({o[0] for o in outs}) == 1, [o[0] for o in outs]
    if shape == "c2":
        assert outs[0][1]["r2_launched"] == 1
    if shape == "aa":                     # sparse Big-path index: the gather variant (k_rank2g)
        assert outs[0][1]["r2_launched"] == 1 and outs[0][1]["r2_units"] > 0.9 * qs.n * 1
    if shape in ("c2_krank", "id90", "small"):
        assert outs[0][1]["r2_launched"] == 0


@pytest.fixture(scope="module")
def aa_small():
    db = synth.make_db(21, 300000, 300, aa=True)
    qs = synth.make_queries(21, db, 20000, 300, aa=True)
    return db, qs


# the two kernels of the sparse Big path: k_rank3g (two filter passes per super-partition, the default) and k_rank2g (UGS_R3=0)
SPARSE_KERNELS = ["1", "0"]


@pytest.mark.parametrize("r3", SPARSE_KERNELS)
@pytest.mark.parametrize("g", ["", "8192", "24576"])
def test_gather_kernel_takes_the_sparse_big_path_and_equals_k_rank(aa_small, g, r3):
    """protein index (rows of tens of postings): k_rank3g / k_rank2g rank every unit, same candidates as the counter kernel; also with
    small partitions (37 of them for 300 k sequences: sub-rows of 0-3 postings, many chunks per unit)"""
    db, qs = aa_small
    a = _search(db, qs, {"UGS_RANK2": "0"}, is_nucleo=False, id=0.8)[0]
    b = _search(db, qs, dict({"UGS_R2_G": g} if g else {}, UGS_R3=r3), is_nucleo=False, id=0.8)[0]
    assert b[1]["r2_kernel"] == ("k_rank3g" if r3 == "1" else "k_rank2g")
    assert a[1]["r2_launched"] == 0 and a[1]["r2_units"] == 0
    assert b[1]["r2_launched"] == 1 and b[1]["r2_units"] > 0.95 * qs.n and b[1]["r2_units"] + b[1]["deferred"] == qs.n
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(x, y)
    assert a[0] == b[0]


@pytest.mark.parametrize("r3", SPARSE_KERNELS)
def test_gather_kernel_short_protein_queries(aa_small, r3):
    """queries of 12-19 residues have <= 15 words (a 4-bit launch of k_rank): the sparse-index kernels take them all the same"""
    db, _ = aa_small
    rng = np.random.default_rng(24)
    rows = db.seqs.reshape(db.n, 300)
    parts, offs = [], [0]
    for _ in range(3000):
        L = int(rng.integers(12, 20)); t = int(rng.integers(0, db.n)); p0 = int(rng.integers(0, 300 - L))
        parts.append(rows[t, p0:p0 + L]); offs.append(offs[-1] + L)
    qs = synth.SeqSet(np.concatenate(parts), np.array(offs, dtype=np.uint64), lambda i: "q%d" % i)
    a = _search(db, qs, {"UGS_RANK2": "0"}, is_nucleo=False, id=0.8)[0]
    b = _search(db, qs, {"UGS_R3": r3}, is_nucleo=False, id=0.8)[0]
    assert (a[1]["rank_kernel"] >> 1) & 0x7f == 4 and b[1]["r2_launched"] == 1 and b[1]["r2_units"] > 0.95 * qs.n
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(x, y)
    assert a[0] == b[0]


@pytest.mark.parametrize("r3", SPARSE_KERNELS)
def test_gather_kernel_deferred_units_and_oracle(aa_small, r3):
    """a kept-key list of 8 entries defers the units with hits to the 8-bit counter kernel behind k_rank3g / k_rank2g; both against the oracle"""
    db, _ = aa_small
    qs = synth.make_queries(22, db, 900, 300, aa=True)
    outs = [_search(db, qs, dict(env, UGS_R3=r3), is_nucleo=False, id=0.8)[0] for env in ({}, {"UGS_R2_KCAP": "8"})]
    assert outs[0][1]["r2_launched"] == 1 and outs[0][1]["deferred"] == 0
    assert outs[1][1]["r2_launched"] == 1 and outs[1][1]["deferred"] > 0 and outs[1][1]["r2_units"] + outs[1][1]["deferred"] == qs.n
    assert outs[0][0] == outs[1][0]
    odb = orc.OrcDB(orc.params(is_nucleo=False, id=0.8), db.seqs, db.offs)
    for out in outs:
        cand, cnt, n = out[2]
        for qi in range(0, qs.n, 9):
            q = qs.seqs[int(qs.offs[qi]):int(qs.offs[qi + 1])]
            on, oc, occ = odb.rank(q, cap=cand.shape[1])
            m = min(on, cand.shape[1])
            assert n[qi] == m and np.array_equal(cand[qi, :m], oc[:m]) and np.array_equal(cnt[qi, :m], occ[:m]), qi


@pytest.mark.parametrize("sp", ["1", "2", "63"])
def test_filter_kernel_super_partition_sizes_and_oracle(aa_small, sp):
    """k_rank3g with super-partitions of one and two partitions of the index (65 536 / 131 072 targets: few false suspects, many
    group-step lists) and starting from ONE super-partition over all 300 k targets (its group-step list overflows: scanned again as
    halves): the same candidates and counts as the oracle's ranking, unit by unit"""
    db, _ = aa_small
    qs = synth.make_queries(25, db, 1500, 300, aa=True)
    out = _search(db, qs, {"UGS_R3": "1", "UGS_R3_SP": sp}, is_nucleo=False, id=0.8)[0]
    kh = out[1]
    assert kh["r2_launched"] == 1 and kh["r2_kernel"] == "k_rank3g" and kh["r2_units"] + kh["deferred"] == qs.n
    assert kh["deferred"] <= qs.n // 100            # (a super-partition that overflows is scanned again as two halves, not deferred)
    odb = orc.OrcDB(orc.params(is_nucleo=False, id=0.8), db.seqs, db.offs)
    cand, cnt, n = out[2]
    for qi in range(0, qs.n, 5):
        q = qs.seqs[int(qs.offs[qi]):int(qs.offs[qi + 1])]
        on, oc, occ = odb.rank(q, cap=cand.shape[1])
        m = min(on, cand.shape[1])
        assert n[qi] == m and np.array_equal(cand[qi, :m], oc[:m]) and np.array_equal(cnt[qi, :m], occ[:m]), qi


def test_filter_kernel_keeps_an_index_with_long_rows(aa_small):
    """an index whose longest rows pass the long-row latch (forced here; a skewed protein dictionary in the field) used to fall back to
    k_rank as a whole: k_rank3g has no sub-row limit and keeps it - same candidates as k_rank's long-row instantiation, nothing deferred"""
    db, qs = aa_small
    a = _search(db, qs, {"UGS_RANK2": "0", "UGS_LONGROWS": "1"}, is_nucleo=False, id=0.8)[0]
    b = _search(db, qs, {"UGS_LONGROWS": "1"}, is_nucleo=False, id=0.8)[0]
    c = _search(db, qs, {"UGS_LONGROWS": "1", "UGS_R3": "0"}, is_nucleo=False, id=0.8)[0]
    assert a[1]["r2_launched"] == 0 and c[1]["r2_launched"] == 0                 # (k_rank2g stays out of long-row indexes)
    assert b[1]["r2_kernel"] == "k_rank3g" and b[1]["r2_units"] + b[1]["deferred"] == qs.n and b[1]["deferred"] <= qs.n // 100
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(x, y)
    assert a[0] == b[0] == c[0]


@pytest.mark.parametrize("r3", SPARSE_KERNELS)
def test_gather_kernel_defers_units_of_abundant_families(r3):
    """families of near-identical sequences give sub-rows far longer than a quad (300 copies: > 255 postings of a row in one partition;
    60 copies: more descriptor lanes than a partition's table holds): those units go to k_rank, the others stay"""
    base = synth.make_db(23, 200000, 300, aa=True)
    rng = np.random.default_rng(23)
    L = 300
    seqs = base.seqs.reshape(base.n, L).copy()
    fams = []
    for start, copies in ((50000, 300), (120000, 60)):
        proto = seqs[start].copy()
        for c in range(copies):
            row = proto.copy()
            pos = rng.integers(0, L, size=6)
            row[pos] = seqs[start + 1000 + c][pos]              # a few substitutions per copy
            seqs[start + c] = row
        fams.append((start, copies))
    db = synth.SeqSet(seqs.reshape(-1), base.offs, lambda i: "t%d" % i)
    qs0 = synth.make_queries(23, db, 3000, 300, aa=True)
    # queries: the synthetic set plus mutated members of both families
    extra = []
    for start, copies in fams:
        for c in range(0, copies, 3):
            row = seqs[start + c].copy()
            pos = rng.integers(0, L, size=20)
            row[pos] = seqs[start + 5000 + c][pos]
            extra.append(row)
    eseq = np.concatenate(extra)
    qseqs = np.concatenate([qs0.seqs, eseq])
    qoffs = np.concatenate([qs0.offs, qs0.offs[-1] + np.arange(1, len(extra) + 1, dtype=np.uint64) * np.uint64(L)])
    qs = synth.SeqSet(qseqs, qoffs, lambda i: "q%d" % i)
    a = _search(db, qs, {"UGS_RANK2": "0", "UGS_LONGROWS": "0"}, is_nucleo=False, id=0.8)[0]
    b = _search(db, qs, {"UGS_LONGROWS": "0", "UGS_R3": r3}, is_nucleo=False, id=0.8)[0]
    assert b[1]["r2_launched"] == 1 and b[1]["deferred"] >= len(extra) // 2 and b[1]["r2_units"] > 2000
    assert b[1]["r2_units"] + b[1]["deferred"] == qs.n
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(x, y)
    assert a[0] == b[0]


def test_small_path_short_queries_with_long_row_kernel():
    """queries of <= 22 letters have <= 15 words: 4-bit counters on the small path; with the long-row switch forced that is the
    instantiation k_rank<SMALL, .., LONG> (found unreached by tests/test_zz_gpu_coverage.py) - same candidates as its twin and the oracle"""
    db = synth.make_db(31, 40000, 250)
    rng = np.random.default_rng(31)
    nq, L = 1500, 22
    src = rng.integers(0, db.n, size=nq); pos = rng.integers(0, 250 - L, size=nq)
    rows = db.seqs.reshape(db.n, 250)
    qseqs = np.stack([rows[s, p:p + L] for s, p in zip(src, pos)]).reshape(-1)
    qs = synth.SeqSet(qseqs, np.arange(nq + 1, dtype=np.uint64) * np.uint64(L), lambda i: "q%d" % i)
    a = _search(db, qs, {"UGS_LONGROWS": "0"}, is_nucleo=True, id=0.97)[0]
    b = _search(db, qs, {"UGS_LONGROWS": "1"}, is_nucleo=True, id=0.97)[0]
    assert a[1]["rank_kernel"] & 1 == 0 and (a[1]["rank_kernel"] >> 1) & 0x7f == 4 and (a[1]["rank_kernel"] >> 9) & 1 == 0
    assert b[1]["rank_kernel"] & 1 == 0 and (b[1]["rank_kernel"] >> 1) & 0x7f == 4 and (b[1]["rank_kernel"] >> 9) & 1 == 1
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(x, y)
    assert a[0] == b[0]
    odb = orc.OrcDB(orc.params(is_nucleo=True, id=0.97), db.seqs, db.offs)
    cand, cnt, n = b[2]
    for qi in range(0, nq, 11):
        on, oc, occ = odb.rank(qs.seqs[qi * L:(qi + 1) * L], cap=cand.shape[1])
        m = min(on, cand.shape[1])
        assert n[qi] == m and np.array_equal(cand[qi, :m], oc[:m]) and np.array_equal(cnt[qi, :m], occ[:m]), qi


def test_packed_query_planes_from_the_setup_kernel(c2_small):
    """UGS_QPK=1: k_rank_setup packs every unit's letters (2 bits + other-letter plane, strand applied) and k_align reads the planes
    instead of packing per unit - same hits, both strands, wildcards in the queries"""
    db, _ = c2_small
    qs = synth.make_queries(17, db, 3000, 250)
    q = qs.seqs.copy()
    rng = np.random.default_rng(17)
    q[rng.integers(0, len(q), size=400)] = ord("N")
    q[rng.integers(0, len(q), size=200)] = ord("R")
    qs = synth.SeqSet(q, qs.offs, lambda i: "q%d" % i)
    a = _search(db, qs, None, is_nucleo=True, id=0.9, strand_both=1)[0]
    b = _search(db, qs, {"UGS_QPK": "1"}, is_nucleo=True, id=0.9, strand_both=1)[0]
    assert a[0] == b[0]


def test_wide_offset_instantiations_of_k_rank(c2_small):
    """an index whose partition table reaches 4 GiB takes the Big-path 4-bit kernels with 64-bit offsets; forced here"""
    db, qs = c2_small
    a = _search(db, qs, {"UGS_RANK2": "0"}, is_nucleo=True, id=0.97)[0]
    b = _search(db, qs, {"UGS_RANK2": "0", "UGS_WIDE_OFFSETS": "1"}, is_nucleo=True, id=0.97)[0]
    c = _search(db, qs, {"UGS_RANK2": "0", "UGS_WIDE_OFFSETS": "1", "UGS_LONGROWS": "1"}, is_nucleo=True, id=0.97)[0]
    assert (a[1]["rank_kernel"] >> 10) & 1 == 0 and (b[1]["rank_kernel"] >> 10) & 1 == 1 and (c[1]["rank_kernel"] >> 9) & 3 == 3
    for x, y, z in zip(a[2], b[2], c[2]):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    assert a[0] == b[0] == c[0]


def _hits_of(db, qs, env, **kw):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k); os.environ[k] = v
    try:
        gdb = capi.UgsDB(capi.params(**kw), db.seqs, db.offs, device=0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
    bat.upload(qs.seqs, qs.offs)
    bat.search(); bat.sync()
    h, nh, pool = bat.fetch()
    st = bat.stats()
    return h, nh, pool, st, bat.kernel_hits()


@pytest.mark.parametrize("kw", [dict(id=0.97), dict(id=0.99, max_accepts=3, max_rejects=8, strand_both=1), dict(id=0.9, max_rejects=5)])
def test_group_filter_of_k_align_equals_the_serial_walk(kw):
    """k_align tests the candidates of a unit four at a time for "no HSP at all" once the unit has rejected one (UGS_ALIGN_GROUP):
    same hit tables, same pair and target-letter counts as the walk that takes every pair alone - families (group members WITH an
    HSP, which go back to the full path), wildcards in targets and queries, targets shorter than two HSP words and longer than 1024"""
    db, qs = synth.make_hard(91, 160, 14, 2500, lmin=40, lmax=1300)
    rng = np.random.default_rng(91)
    d = db.seqs.copy()
    for t in rng.integers(0, db.n, size=300):                       # wildcards in 300 targets
        d[int(db.offs[t]) + int(rng.integers(0, int(db.offs[t + 1] - db.offs[t])))] = ord("N")
    lens = np.diff(db.offs.astype(np.int64))
    short = np.array([7, 9, 10, 12], dtype=np.int64)                # around 2 x the HSP word length
    seqs = np.concatenate([d, synth._random_letters(rng, int(short.sum()), synth.NT)])
    offs = np.concatenate([[0], np.cumsum(np.concatenate([lens, short]))]).astype(np.uint64)
    db = synth.SeqSet(seqs, offs, lambda i: "t%d" % i)
    rq = synth.make_db(92, 400, 250)                                # + random queries: every candidate a reject
    qseqs = np.concatenate([qs.seqs, rq.seqs])
    qoffs = np.concatenate([qs.offs, rq.offs[1:] + qs.offs[-1]]).astype(np.uint64)
    qs = synth.SeqSet(qseqs, qoffs, lambda i: "q%d" % i)
    ref = _hits_of(db, qs, {"UGS_ALIGN_GROUP": "0"}, is_nucleo=True, **kw)
    assert ref[4]["group_rejects"] == 0 and len(ref[0]) > 200
    for g in (None, "2", "3"):
        got = _hits_of(db, qs, None if g is None else {"UGS_ALIGN_GROUP": g}, is_nucleo=True, **kw)
        assert got[4]["group_rejects"] > 0, g
        assert np.array_equal(got[1], ref[1]), g
        for f in ("query", "target", "ids", "mism", "gaps_int", "aln_len", "opens", "qlo", "qhi", "tlo", "thi", "ql", "tl", "strand", "cigar_len", "cols"):
            assert np.array_equal(got[0][f], ref[0][f]), (g, f)
        for a, b in zip(got[0][:2000], ref[0][:2000]):                  # (the run pool is handed out in the order the waves finish)
            assert np.array_equal(got[2][int(a["cigar_off"]):int(a["cigar_off"]) + int(a["cigar_len"])], ref[2][int(b["cigar_off"]):int(b["cigar_off"]) + int(b["cigar_len"])]), g
        assert got[3]["pairs_aligned"] == ref[3]["pairs_aligned"] and got[3]["target_letters"] == ref[3]["target_letters"], g


@pytest.mark.parametrize("kw", [dict(id=0.8), dict(id=0.9, max_accepts=2, max_rejects=6)])
def test_group_filter_of_k_align_amino_acid_targets(kw):
    """the same for a protein database (targets kept as bytes, bucketed word table, BLOSUM extension): UGS_ALIGN_GROUP 0 / 1 / 2 give the
    same hit tables and counters - families, wildcards (X, B, Z) in targets and queries, targets around two HSP words and beyond 1024"""
    db, qs = synth.make_hard(93, 120, 12, 1800, lmin=40, lmax=1200, aa=True)
    rng = np.random.default_rng(93)
    d = db.seqs.copy()
    for t in rng.integers(0, db.n, size=200):
        d[int(db.offs[t]) + int(rng.integers(0, int(db.offs[t + 1] - db.offs[t])))] = ord("X")
    lens = np.diff(db.offs.astype(np.int64))
    short = np.array([4, 5, 6, 7, 9], dtype=np.int64)
    seqs = np.concatenate([d, synth._random_letters(rng, int(short.sum()), synth.AA, synth.RR_FREQ)])
    offs = np.concatenate([[0], np.cumsum(np.concatenate([lens, short]))]).astype(np.uint64)
    db = synth.SeqSet(seqs, offs, lambda i: "t%d" % i)
    rq = synth.make_db(94, 400, 300, aa=True)
    qseqs = np.concatenate([qs.seqs, rq.seqs])
    qoffs = np.concatenate([qs.offs, rq.offs[1:] + qs.offs[-1]]).astype(np.uint64)
    qs = synth.SeqSet(qseqs, qoffs, lambda i: "q%d" % i)
    ref = _hits_of(db, qs, {"UGS_ALIGN_GROUP": "0"}, is_nucleo=False, **kw)
    assert ref[4]["group_rejects"] == 0 and len(ref[0]) > 200
    for g in (None, "2"):
        got = _hits_of(db, qs, None if g is None else {"UGS_ALIGN_GROUP": g}, is_nucleo=False, **kw)
        assert got[4]["group_rejects"] > 0, g
        assert np.array_equal(got[1], ref[1]), g
        for f in ("query", "target", "ids", "mism", "gaps_int", "aln_len", "opens", "qlo", "qhi", "tlo", "thi", "ql", "tl", "strand", "cigar_len", "cols"):
            assert np.array_equal(got[0][f], ref[0][f]), (g, f)
        for a, b in zip(got[0][:2000], ref[0][:2000]):
            assert np.array_equal(got[2][int(a["cigar_off"]):int(a["cigar_off"]) + int(a["cigar_len"])], ref[2][int(b["cigar_off"]):int(b["cigar_off"]) + int(b["cigar_len"])]), g
        assert got[3]["pairs_aligned"] == ref[3]["pairs_aligned"] and got[3]["target_letters"] == ref[3]["target_letters"], g
