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
    assert np.all(hits["flags"] & 1 == 1)
    # AlignResult::GetRawScore rescoring the path gives the x-drop score back, and local paths start and end on M
    assert orc.lib().orc_local_rescore_diffs() == before
    assert np.all(hits["cols"] == hits["aln_len"])


def test_local_evalue_formula():
    p = orc.params(True, id=None, local_evalue=1e-6)
    e, b = C.c_double(0), C.c_double(0)
    orc.lib().orc_local_evalue(C.byref(p), 302.0, 344, C.byref(e), C.byref(b))     # first line of the hard_small probe
    assert "%.1f" % b.value == "558.8" and "%.2g" % e.value == "2.1e-157"


@pytest.mark.parametrize("name", [n for n in G.local_case_names() if G.LOCAL_MANIFEST[n].get("userout")])
def test_local_userout_from_oracle_hits(name):
    """ugs_format_userout_local (host C++) fed with the oracle's local hits reproduces the reference's -userout with every
    supported field (sha256 of the whole file + its first lines): HSP coordinates, segments, coverages, evalue, raw, bits"""
    import hashlib
    import os
    from usearch12_amd import capi
    c, db, qs, b6 = G.load_local(name)
    u = G.LOCAL_MANIFEST[name]["userout"]
    p = orc.params(is_nucleo=not c["aa"], **G.local_params_kw(c))
    odb = orc.OrcDB(p, db.seqs, db.offs)
    hits, nh, pool = odb.search(qs.seqs, qs.offs, nthreads=4)
    masked = odb.masked()
    L = capi.lib()
    L.ugs_format_userout_local.restype = C.c_int
    L.ugs_format_userout_local.argtypes = [C.POINTER(type(p)), C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint32,
                                           C.c_void_p, C.c_uint32, C.c_char_p, C.c_int]
    pool = np.ascontiguousarray(pool, np.uint32)
    buf = C.create_string_buffer(1 << 16)
    out, k = [], 0
    ql, tl = qs.labels(), db.labels()
    for qi in range(qs.n):
        q = np.ascontiguousarray(qs.seqs[int(qs.offs[qi]):int(qs.offs[qi + 1])])
        for j in range(int(nh[qi])):
            h = hits[k:k + 1]; k += 1
            t = int(h["target"][0])
            ts = np.ascontiguousarray(masked[int(db.offs[t]):int(db.offs[t + 1])])
            n = L.ugs_format_userout_local(C.byref(p), h.ctypes.data, pool.ctypes.data, u["fields"].encode(), ql[qi].encode(), tl[t].encode(),
                                           q.ctypes.data, len(q), ts.ctypes.data, len(ts), buf, len(buf))
            assert 0 < n < len(buf), capi.last_error()
            out.append(buf.raw[:n])
    got = b"".join(out)
    head = open(os.path.join(G.GOLD, name + ".user.head"), "rb").read()
    assert got[:len(head)] == head
    assert got.count(b"\n") == u["lines"] and hashlib.sha256(got).hexdigest() == u["sha256"]


@pytest.mark.parametrize("name", [n for n in G.local_case_names() if G.LOCAL_MANIFEST[n].get("userout")])
def test_local_alnout_from_oracle_hits(name):
    """-alnout of usearch_local (hit table with scores / e-values / segments, alignment rows over the HSP, summary with
    score, bits, e-value) from the oracle's hits: sha256 of the reference's whole file + its first 60 lines"""
    import hashlib
    import os
    from usearch12_amd import capi
    c, db, qs, b6 = G.load_local(name)
    u = G.LOCAL_MANIFEST[name]["userout"]
    p = orc.params(is_nucleo=not c["aa"], **G.local_params_kw(c))
    odb = orc.OrcDB(p, db.seqs, db.offs)
    hits, nh, pool = odb.search(qs.seqs, qs.offs, nthreads=4)
    masked = odb.masked()
    L = capi.lib()
    P = C.POINTER(type(p))
    L.ugs_format_alnout_header_local.argtypes = [P, C.c_void_p, C.c_uint32, C.c_char_p, C.POINTER(C.c_char_p), C.c_char_p, C.c_int]
    L.ugs_format_alnout_hit_local.argtypes = [P, C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_char_p, C.c_int]
    pool = np.ascontiguousarray(pool, np.uint32)
    buf = C.create_string_buffer(1 << 18)
    out, k = [], 0
    ql, tl = qs.labels(), db.labels()
    for qi in range(qs.n):
        n = int(nh[qi])
        if n == 0:
            continue
        h = hits[k:k + n]
        labs = (C.c_char_p * n)(*[tl[int(t)].encode() for t in h["target"]])
        m = L.ugs_format_alnout_header_local(C.byref(p), h.ctypes.data, n, ql[qi].encode(), labs, buf, len(buf))
        assert 0 < m < len(buf)
        out.append(buf.raw[:m])
        q = np.ascontiguousarray(qs.seqs[int(qs.offs[qi]):int(qs.offs[qi + 1])])
        for j in range(n):
            hj = hits[k + j:k + j + 1]
            t = int(hj["target"][0])
            ts = np.ascontiguousarray(masked[int(db.offs[t]):int(db.offs[t + 1])])
            m = L.ugs_format_alnout_hit_local(C.byref(p), hj.ctypes.data, pool.ctypes.data, ql[qi].encode(), tl[t].encode(), q.ctypes.data, len(q),
                                              ts.ctypes.data, len(ts), buf, len(buf))
            assert 0 < m < len(buf), capi.last_error()
            out.append(buf.raw[:m])
        k += n
    got = b"".join(out)
    head = open(os.path.join(G.GOLD, name + ".aln.head"), "rb").read()
    assert got[:len(head)] == head
    assert got.count(b"\n") == u["aln_lines"] and hashlib.sha256(got).hexdigest() == u["aln_sha256"]


@pytest.mark.parametrize("name", [n for n in G.local_case_names() if G.LOCAL_MANIFEST[n].get("userout")])
def test_local_uc_and_hit_count_rules_from_oracle_hits(name):
    """-uc of usearch_local (HSP start columns) and -top_hits_only / -top_hit_only on local scores (ugs_hits_to_report),
    fed with the oracle's hits: sha256 of the reference's files"""
    import hashlib
    from usearch12_amd import capi
    c, db, qs, b6 = G.load_local(name)
    u = G.LOCAL_MANIFEST[name]["userout"]
    p = orc.params(is_nucleo=not c["aa"], **G.local_params_kw(c))
    hits, nh, pool = orc.OrcDB(p, db.seqs, db.offs).search(qs.seqs, qs.offs, nthreads=4)
    L = capi.lib()
    L.ugs_hits_to_report.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    L.ugs_hits_to_report.restype = C.c_uint32
    pool = np.ascontiguousarray(pool, np.uint32)
    buf = C.create_string_buffer(1 << 16)
    ql, tl = qs.labels(), db.labels()
    qlens = np.diff(qs.offs.astype(np.int64))
    uc, tops, top1 = [], [], []
    k = 0
    for qi in range(qs.n):
        n = int(nh[qi])
        h = hits[k:k + n]
        if n == 0:
            L.ugs_format_uc_nohit(int(qlens[qi]), ql[qi].encode(), buf, len(buf)); uc.append(buf.value)
        for j in range(n):
            L.ugs_format_uc_hit(h[j:j + 1].ctypes.data, pool.ctypes.data, 0 if c["aa"] else 1, ql[qi].encode(), tl[int(h["target"][j])].encode(), buf, len(buf))
            uc.append(buf.value)
        if n:
            first = C.c_uint32(0)
            m = L.ugs_hits_to_report(h.ctypes.data, n, 0, 0, 1, C.byref(first))
            for j in range(m):
                L.ugs_format_blast6_local(C.byref(p), h[first.value + j:first.value + j + 1].ctypes.data, ql[qi].encode(),
                                          tl[int(h["target"][first.value + j])].encode(), buf, len(buf))
                tops.append(buf.value)
            m = L.ugs_hits_to_report(h.ctypes.data, n, 0, 1, 0, C.byref(first))
            assert m == 1
            L.ugs_format_blast6_local(C.byref(p), h[first.value:first.value + 1].ctypes.data, ql[qi].encode(), tl[int(h["target"][first.value])].encode(), buf, len(buf))
            top1.append(buf.value)
        k += n
    for got, key in ((b"".join(uc), "uc"), (b"".join(tops), "top_hits_only"), (b"".join(top1), "top_hit_only")):
        assert got.count(b"\n") == u[key + "_lines"], key
        assert hashlib.sha256(got).hexdigest() == u[key + "_sha256"], key
