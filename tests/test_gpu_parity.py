"""GPU parity tests proper (-m gpu): the HIP path through the C-ABI vs (a) the reference's
golden text and (b) the oracle, bit-exact on hit identity, coordinates, paths and text."""
import os

import numpy as np
import pytest

import golden_util as G
import orc
from usearch12_amd import capi

pytestmark = pytest.mark.gpu

# Round-1 device scope: the Big ranking path (DB > -big) for nt and aa; the small path is
# implemented too and covered by the same cases.
CASES = G.case_names()


def _run_gpu(c, db, qs):
    p = capi.params(is_nucleo=not c["aa"], id=c["id"], **G.params_kw(c))
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    hits, nh, pool = gdb.search(qs.seqs, qs.offs)
    return p, gdb, hits, nh, pool


@pytest.mark.parametrize("name", CASES)
def test_gpu_matches_reference_text(name):
    c, db, qs, b6, uc = G.load(name)
    p, gdb, hits, nh, pool = _run_gpu(c, db, qs)
    qlens = np.diff(qs.offs.astype(np.int64))
    gb6, guc = orc.format_outputs(capi.lib(), "ugs", hits, nh, pool, qs.labels(), qlens, db.labels(), not c["aa"])
    assert gb6 == b6
    assert guc == uc


@pytest.mark.parametrize("name", ["nt_big", "hard_big", "hard_small", "hard_aa", "hard_acc"])
def test_gpu_hits_equal_oracle_records(name):
    c, db, qs, b6, uc = G.load(name)
    p, gdb, hits, nh, pool = _run_gpu(c, db, qs)
    op = orc.params(is_nucleo=not c["aa"], id=c["id"], **G.params_kw(c))
    odb = orc.OrcDB(op, db.seqs, db.offs)
    ohits, onh, opool = odb.search(qs.seqs, qs.offs, nthreads=4)
    assert np.array_equal(nh, onh)
    for f in hits.dtype.names:
        if f != "cigar_off":          # pool layout is an implementation detail; runs compared below
            assert np.array_equal(hits[f], ohits[f]), f
    for h, o in zip(hits, ohits):
        assert np.array_equal(pool[int(h["cigar_off"]):int(h["cigar_off"]) + int(h["cigar_len"])],
                              opool[int(o["cigar_off"]):int(o["cigar_off"]) + int(o["cigar_len"])])


@pytest.mark.parametrize("name", ["nt_big", "hard_big", "hard_id90", "hard_small", "nt_small", "hard_aa", "aa_small"])
def test_gpu_candidate_order_equals_oracle(name):
    c, db, qs, b6, uc = G.load(name)
    p = capi.params(is_nucleo=not c["aa"], id=c["id"], **G.params_kw(c))
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
    bat.upload(qs.seqs, qs.offs)
    bat.search(); bat.sync()
    cand, cnt, n = bat.candidates()
    op = orc.params(is_nucleo=not c["aa"], id=c["id"], **G.params_kw(c))
    odb = orc.OrcDB(op, db.seqs, db.offs)
    K = cand.shape[1]
    ns = 2 if p.strand_both else 1
    for qi in range(0, qs.n, 7):
        q = qs.seqs[int(qs.offs[qi]):int(qs.offs[qi + 1])]
        for s in range(ns):
            qq = q if s == 0 else orc.revcomp(q)
            on, ocand, ocnt = odb.rank(qq, cap=K)
            u = qi * ns + s
            m = min(on, K)
            assert n[u] == m, (qi, s, n[u], on)
            assert np.array_equal(cand[u, :m], ocand[:m]), (qi, s)
            assert np.array_equal(cnt[u, :m], ocnt[:m]), (qi, s)


def test_gpu_mask_and_index_equal_oracle():
    c, db, qs, b6, uc = G.load("hard_small")
    p = capi.params(is_nucleo=True, id=c["id"])
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    masked, row_off, postings = gdb.debug_fetch()
    odb = orc.OrcDB(orc.params(is_nucleo=True, id=c["id"]), db.seqs, db.offs)
    assert np.array_equal(masked, odb.masked())
    oro, opo = odb.index()
    assert np.array_equal(row_off, oro)
    assert np.array_equal(postings, opo)


def test_gpu_small_partitions(monkeypatch):
    """Force tiny target partitions (many per DB) so the multi-partition scan, the per-partition
    count-1 fill quota and cross-partition tie order are all exercised on a small DB."""
    monkeypatch.setenv("UGS_GSHIFT", "7")
    for name in ("hard_big", "nt_small", "hard_both"):
        c, db, qs, b6, uc = G.load(name)
        p, gdb, hits, nh, pool = _run_gpu(c, db, qs)
        qlens = np.diff(qs.offs.astype(np.int64))
        gb6, guc = orc.format_outputs(capi.lib(), "ugs", hits, nh, pool, qs.labels(), qlens, db.labels(), True)
        assert gb6 == b6, name
        assert guc == uc, name


def test_gpu_device_results_view_matches_fetch():
    """The device-resident hit table exposed for the RCCL gather (ugs_batch_device_results) holds the
    same records ugs_batch_fetch returns."""
    from conftest import d2h
    from usearch12_amd import multigpu
    c, db, qs, b6, uc = G.load("hard_acc")
    p = capi.params(is_nucleo=True, id=c["id"], **G.params_kw(c))
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
    bat.upload(qs.seqs, qs.offs); bat.search(); bat.sync()
    hits, nh, pool = bat.fetch()
    (ph, bh), (pn, bn), (pc, bc) = bat.device_results(query_base=1000)
    t_h, t_n, t_c = d2h(ph, bh), d2h(pn, bn), d2h(pc, bc)
    ghits, gcnt, gpool = multigpu.merge_tables([t_h], [t_n], [t_c])
    assert len(ghits) == len(hits) and np.array_equal(gcnt, nh)
    ghits["query"] -= 1000
    # fetch() re-sorts the hits of a query by score; compare as per-query sets
    key = lambda a: sorted(zip(a["query"].tolist(), a["target"].tolist(), a["ids"].tolist(), a["aln_len"].tolist()))
    assert key(ghits) == key(hits)


def test_cli_text_identical_to_reference(tmp_path):
    """The C++ driver (usearch12_amd/ugs_cli, the reference's command line for this one command):
    FASTA in, -blast6out / -uc out, byte-identical to the reference's files."""
    import subprocess
    cli = os.path.join(os.path.dirname(capi.LIB_PATH), "ugs_cli")
    for name in ("hard_both", "hard_aa", "hard_filt", "hard_filt_s", "hard_fulldp", "hard_gaforce", "hard_hardmask", "hard_termid", "hard_termidd",
                 "hard_noid", "hard_noid_s", "deep_all_s", "deep_acc100", "deep_rej256", "deep_aa",          # (deep_*: walks past 64 candidates, > 64 hits per query)
                 "hard_band0", "opt_word6", "opt_word7_s", "opt_word4_aa", "opt_step3_s", "opt_bump90_s", "opt_minhsp", "opt_match", "opt_hspw4", "opt_hspw2_aa", "opt_mask_none", "opt_mask_user_aa"):
        c, db, qs, b6, uc = G.load(name)
        dbfa, qfa = str(tmp_path / "db.fa"), str(tmp_path / "q.fa")
        db.write_fasta(dbfa); qs.write_fasta(qfa)
        cmd = [cli, "-usearch_global", qfa, "-db", dbfa] + (["-id", str(c["id"])] if c["id"] is not None else []) + \
              ["-blast6out", str(tmp_path / "o.b6"), "-uc", str(tmp_path / "o.uc"), "-batch", "500"]
        if not c["aa"]:
            cmd += ["-strand", c["strand"]]
        for opt in ("big", "maxaccepts", "maxrejects") + G._mg.FILTER_OPTS:
            if opt in c:
                cmd += ["-" + opt, str(c[opt])]
        cmd += ["-" + f for f in ("fulldp", "gaforce", "hardmask") if c.get(f)]
        for opt in ("termid", "termidd", "band") + G._mg.EXTRA_OPTS:
            if opt in c:
                cmd += ["-" + opt, str(c[opt])]
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        assert open(tmp_path / "o.b6").read() == b6, name
        assert open(tmp_path / "o.uc").read() == uc, name


@pytest.mark.parametrize("shape", ["aa_60k", "nt_both_120k", "nt_long_1500", "nt_small_60k"])
def test_gpu_medium_shapes_equal_oracle(shape):
    """Shapes beyond the golden cases (real Big path on >100k targets, both strands, 1500-letter
    sequences, protein, small path at tens of thousands of targets): every hit record and path
    equals the oracle's."""
    from usearch12_amd import synth
    kw, aa, ident = {}, False, 0.97
    if shape == "aa_60k":
        aa, ident = True, 0.8
        db = synth.make_db(5, 60_000, 300, aa=True); qs = synth.make_queries(5, db, 8_000, 300, aa=True)
        kw = dict(big=1000)
    elif shape == "nt_both_120k":
        db = synth.make_db(31, 120_000, 250); qs = synth.revcomp_some(31, synth.make_queries(31, db, 8_000, 250))
        kw = dict(strand_both=1)
    elif shape == "nt_long_1500":
        db, qs = synth.make_hard(32, 600, 10, 800, lmin=1200, lmax=1600)
        kw = dict(big=100)
    else:
        db = synth.make_db(34, 60_000, 250); qs = synth.make_queries(34, db, 4_000, 250)
    p = capi.params(is_nucleo=not aa, id=ident, **kw)
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    hits, nh, pool = gdb.search(qs.seqs, qs.offs)
    odb = orc.OrcDB(orc.params(is_nucleo=not aa, id=ident, **kw), db.seqs, db.offs)
    oh, onh, opool = odb.search(qs.seqs, qs.offs, nthreads=min(32, os.cpu_count() or 1))
    assert np.array_equal(nh, onh)
    assert len(hits) > 0.5 * qs.n
    for f in hits.dtype.names:
        if f != "cigar_off":
            assert np.array_equal(hits[f], oh[f]), f
    for a, b in zip(hits[::53], oh[::53]):
        assert np.array_equal(pool[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["cigar_len"])],
                              opool[int(b["cigar_off"]):int(b["cigar_off"]) + int(b["cigar_len"])])


def test_gpu_full_size_properties():
    """BASELINE.json configs[1] at FULL size (1M x 250 nt vs 1M-seq DB) through size-independent
    properties: per-hit invariants of the reference semantics, path/length consistency, ranking
    order, determinism (two runs bit-identical) and agreement with the oracle on a sample."""
    from usearch12_amd import synth
    db = synth.make_db(2, 1_000_000, 250)
    qs = synth.make_queries(2, db, 1_000_000, 250)
    p = capi.params(is_nucleo=True, id=0.97)
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
    bat.upload(qs.seqs, qs.offs)
    bat.search(); bat.sync()
    hits, nh, pool = bat.fetch()
    cand, cnt, n = bat.candidates()
    # accept rule and bookkeeping (accepter.cpp:35-39, terminator.cpp:64-100, arscorer.cpp:201-296)
    assert nh.max() <= 1 and len(hits) == int(nh.sum())
    assert np.all(hits["ids"].astype(np.float64) / hits["aln_len"] >= float(np.float32(0.97)))
    assert np.all(hits["ids"] + hits["mism"] + hits["gaps_int"] == hits["aln_len"])
    assert np.array_equal(hits["query"], np.flatnonzero(nh))
    qlen = np.diff(qs.offs.astype(np.int64))
    assert np.array_equal(hits["ql"], qlen[hits["query"]]) and np.all(hits["tl"] == 250)
    # path consistency: M+D columns == query length, M+I columns == target length, cols == sum of runs
    starts = hits["cigar_off"].astype(np.int64); lens = hits["cigar_len"].astype(np.int64)
    order = np.argsort(starts, kind="stable")       # pool order is allocation order on the device
    run_hit = np.repeat(np.arange(len(hits))[order], lens[order])
    assert int(lens.sum()) == len(pool)
    runs = pool[np.concatenate([np.arange(s, s + l) for s, l in zip(starts[order][:2000], lens[order][:2000])])]
    rh = run_hit[:len(runs)]
    op, ln = runs & 3, (runs >> 2).astype(np.int64)
    for h in np.unique(rh)[:2000]:
        m = rh == h
        assert ln[m][op[m] != 2].sum() == hits["ql"][h] and ln[m][op[m] != 1].sum() == hits["tl"][h]
        assert ln[m].sum() == hits["cols"][h]
    # true source recovered for the overwhelming majority of mutated queries
    src = qs.src[hits["query"]]
    assert np.mean(src == hits["target"].astype(np.int64)) > 0.999
    assert 0.80 < len(hits) / qs.n < 0.83
    # ranking order: counts never increase along a candidate list (count sort, countsort.cpp:110-191)
    c = cnt.astype(np.int64)
    valid = np.arange(c.shape[1])[None, :] < n[:, None]
    assert np.all((np.diff(c, axis=1) <= 0) | ~valid[:, 1:])
    # determinism
    bat.search(); bat.sync()
    hits2, nh2, pool2 = bat.fetch()
    assert np.array_equal(nh, nh2)
    for f in hits.dtype.names:
        if f != "cigar_off":
            assert np.array_equal(hits[f], hits2[f]), f
    # oracle agreement on a 3000-query sample of the same batch (full DB)
    sample = qs.slice(500_000, 503_000)
    odb = orc.OrcDB(orc.params(is_nucleo=True, id=0.97), db.seqs, db.offs)
    oh, onh, opool = odb.search(sample.seqs, sample.offs, nthreads=min(32, os.cpu_count() or 1))
    sel = (hits["query"] >= 500_000) & (hits["query"] < 503_000)
    assert np.array_equal(nh[500_000:503_000], onh)
    for f in ("target", "ids", "mism", "gaps_int", "aln_len", "opens", "qlo", "qhi", "tlo", "thi", "cigar_len", "cols"):
        assert np.array_equal(hits[f][sel], oh[f]), f


def test_gpu_fetch_into_reused_pinned_buffers():
    """fetch(reuse=True): page-locked result buffers owned by the batch (grown on UGS_E_CAPACITY) hold the same records"""
    c, db, qs, b6, uc = G.load("hard_acc")
    p = capi.params(is_nucleo=True, id=c["id"], **G.params_kw(c))
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
    bat.upload(qs.seqs, qs.offs); bat.search(); bat.sync()
    h0, n0, p0 = bat.fetch()
    for _ in range(2):
        h1, n1, p1 = bat.fetch(reuse=True)
        assert np.array_equal(n0, n1)
        for f in h0.dtype.names:
            if f != "cigar_off":                      # the pool is filled in allocation order, which differs between searches
                assert np.array_equal(h0[f], h1[f]), f
        for a, b in zip(h0, h1):
            assert np.array_equal(p0[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["cigar_len"])],
                                  p1[int(b["cigar_off"]):int(b["cigar_off"]) + int(b["cigar_len"])])
        bat.search(); bat.sync()
    bat.close()
