"""Writers beyond blast6/uc: -userout/-userfields, -output_no_hits, -matched/-notmatched,
-dbmatched/-dbnotmatched, -maxhits/-top_hits_only/-top_hit_only.  Expected text: the reference's own files
(tests/golden/make_golden_outputs.py; big ones as sha256 + first lines).

CPU part: the host-side formatting functions of libugs.so fed with the ORACLE's hit records reproduce the
reference's files (formatting only; no search runs in the product library here).
GPU part: the CLI driver end to end."""
import ctypes as C
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import golden_util as G
import orc
from usearch12_amd import capi

MAN = json.load(open(os.path.join(G.GOLD, "out_manifest.json")))


def check(run, kind, text):
    info = MAN[run]["files"][kind]
    data = text if isinstance(text, bytes) else text.encode()
    if info["whole"]:
        want = open(os.path.join(G.GOLD, "out_%s.%s" % (run, kind)), "rb").read()
        assert data == want, (run, kind)
    else:
        head = open(os.path.join(G.GOLD, "out_%s.%s.head" % (run, kind)), "rb").read()
        assert data[:len(head)] == head, (run, kind, "head")
        assert data.count(b"\n") == info["lines"] and hashlib.sha256(data).hexdigest() == info["sha256"], (run, kind)


def opt(extra, name, default=None):
    return extra[extra.index(name) + 1] if name in extra else default


def format_run(run):
    """all output files of one run from oracle hits + libugs writers -> dict kind -> text"""
    L = capi.lib()
    m = MAN[run]
    c, db, qs, _, _ = G.load(m["case"])
    extra = m["extra"]
    fields = (opt(extra, "-userfields") or "query").encode()
    maxhits = int(opt(extra, "-maxhits", 0))
    top1, tops, nohits = "-top_hit_only" in extra, "-top_hits_only" in extra, "-output_no_hits" in extra
    nucleo = not c["aa"]
    p = orc.params(is_nucleo=nucleo, id=c["id"], **G.params_kw(c))
    odb = orc.OrcDB(p, db.seqs, db.offs)
    hits, nh, pool = odb.search(qs.seqs, qs.offs, nthreads=4)
    masked = odb.masked().tobytes()
    qlabels, tlabels = qs.labels(), db.labels()
    buf = C.create_string_buffer(1 << 20)
    out = {k: [] for k in ("user", "b6", "uc", "matched", "notmatched", "dbmatched", "dbnotmatched", "aln", "pairs", "qseg", "tseg", "trim")}
    dbcount = np.zeros(db.n, np.int64)
    L.ugs_format_userout.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32,
                                     C.c_char_p, C.c_uint32, C.c_char_p, C.c_int]
    L.ugs_hits_to_report.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    L.ugs_hits_to_report.restype = C.c_uint32
    L.ugs_format_fasta.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_char_p, C.c_int]
    L.ugs_format_blast6_nohit.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.ugs_format_fastapairs.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_char_p, C.c_int]
    L.ugs_format_trimout.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_char_p, C.c_int]
    L.ugs_format_segout.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_char_p, C.c_int]
    L.ugs_format_alnout_header.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.POINTER(C.c_char_p), C.c_char_p, C.c_int]
    L.ugs_format_alnout_hit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32,
                                        C.c_char_p, C.c_int]
    qbytes, k = qs.seqs.tobytes(), 0

    def take(n):
        assert 0 <= n < len(buf), (n, capi.lib().ugs_last_error())
        return buf.raw[:n]
    for qi in range(qs.n):
        qseq = qbytes[int(qs.offs[qi]):int(qs.offs[qi + 1])]
        ql, qlab = len(qseq), qlabels[qi].encode()
        n_all = int(nh[qi])
        h = hits[k:k + n_all]
        k += n_all
        first = C.c_uint32(0)
        n = L.ugs_hits_to_report(h.ctypes.data, n_all, maxhits, int(top1), int(tops), C.byref(first)) if n_all else 0
        h = h[first.value:]
        if n == 0:
            out["uc"].append(take(L.ugs_format_uc_nohit(ql, qlab, buf, len(buf))))
            if nohits:
                out["b6"].append(take(L.ugs_format_blast6_nohit(qlab, buf, len(buf))))
                out["user"].append(take(L.ugs_format_userout(None, None, int(nucleo), fields, qlab, None, qseq, ql, None, 0, buf, len(buf))))
            out["notmatched"].append(take(L.ugs_format_fasta(qlab, qseq, ql, buf, len(buf))))
            continue
        tl_arr = (C.c_char_p * n)(*[tlabels[int(h[j]["target"])].encode() for j in range(n)])
        out["aln"].append(take(L.ugs_format_alnout_header(h.ctypes.data, n, qlab, tl_arr, buf, len(buf))))
        for j in range(n):
            t = int(h[j]["target"])
            tlab = tlabels[t].encode()
            tseq = masked[int(db.offs[t]):int(db.offs[t + 1])]
            hp = h[j:j + 1].ctypes.data
            out["b6"].append(take(L.ugs_format_blast6(hp, qlab, tlab, buf, len(buf))))
            out["uc"].append(take(L.ugs_format_uc_hit(hp, pool.ctypes.data, int(nucleo), qlab, tlab, buf, len(buf))))
            out["user"].append(take(L.ugs_format_userout(hp, pool.ctypes.data, int(nucleo), fields, qlab, tlab, qseq, ql, tseq, len(tseq), buf, len(buf))))
            out["pairs"].append(take(L.ugs_format_fastapairs(hp, pool.ctypes.data, qlab, tlab, qseq, ql, tseq, len(tseq), buf, len(buf))))
            out["trim"].append(take(L.ugs_format_trimout(hp, pool.ctypes.data, qlab, qseq, ql, buf, len(buf))))
            out["qseg"].append(take(L.ugs_format_segout(hp, pool.ctypes.data, 0, qlab, tlab, qseq, ql, tseq, len(tseq), buf, len(buf))))
            out["tseg"].append(take(L.ugs_format_segout(hp, pool.ctypes.data, 1, qlab, tlab, qseq, ql, tseq, len(tseq), buf, len(buf))))
            out["aln"].append(take(L.ugs_format_alnout_hit(hp, pool.ctypes.data, int(nucleo), qlab, tlab, qseq, ql, tseq, len(tseq), buf, len(buf))))
            dbcount[t] += 1
        out["matched"].append(take(L.ugs_format_fasta(qlab, qseq, ql, buf, len(buf))))
    for t in range(db.n):
        tseq = masked[int(db.offs[t]):int(db.offs[t + 1])]
        rec = take(L.ugs_format_fasta(tlabels[t].encode(), tseq, len(tseq), buf, len(buf)))
        out["dbmatched" if dbcount[t] else "dbnotmatched"].append(rec)
    return {k: b"".join(v) for k, v in out.items()}


@pytest.mark.parametrize("run", sorted(MAN))
def test_writers_reproduce_reference_files(run):
    out = format_run(run)
    for kind in MAN[run]["files"]:
        check(run, kind, out[kind])


def test_unknown_userfield_is_an_error():
    assert capi.lib().ugs_userfields_check(b"query+nosuchfield") < 0
    assert capi.lib().ugs_userfields_check(b"query+target+id") == 0


@pytest.mark.gpu
@pytest.mark.parametrize("run", sorted(MAN))
def test_cli_outputs_identical_to_reference(tmp_path, run):
    m = MAN[run]
    c, db, qs, _, _ = G.load(m["case"])
    tmp = str(tmp_path)
    dbfa, qfa = os.path.join(tmp, "db.fa"), os.path.join(tmp, "q.fa")
    db.write_fasta(dbfa)
    qs.write_fasta(qfa)
    cli = os.path.join(os.path.dirname(capi.LIB_PATH), "ugs_cli")
    cmd = [cli, "-usearch_global", qfa, "-db", dbfa, "-id", str(c["id"])]
    if not c["aa"]:
        cmd += ["-strand", c["strand"]]
    for o in ("big", "maxaccepts", "maxrejects"):
        if o in c:
            cmd += ["-" + o, str(c[o])]
    cmd += m["extra"]
    names = {"trim": "-trimout", "pairs": "-fastapairs", "qseg": "-qsegout", "tseg": "-tsegout", "aln": "-alnout", "user": "-userout", "b6": "-blast6out", "uc": "-uc", "matched": "-matched", "notmatched": "-notmatched",
             "dbmatched": "-dbmatched", "dbnotmatched": "-dbnotmatched"}
    for kind in m["files"]:
        cmd += [names[kind], os.path.join(tmp, "o." + kind)]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    for kind in m["files"]:
        check(run, kind, open(os.path.join(tmp, "o." + kind), "rb").read())


@pytest.mark.gpu
def test_cli_fastq_queries_identical_to_reference(tmp_path):
    """FASTQ query file (FASTQSeqSource fastqseqsource.cpp:7-107): same hits as the FASTA run, -matchedfq / -notmatchedfq
    (SeqToFastq seqdb.cpp:14-29) byte-identical to the reference's files"""
    import hashlib
    import importlib.util
    spec = importlib.util.spec_from_file_location("mgf", os.path.join(G.GOLD, "make_golden_fastq.py"))
    mgf = importlib.util.module_from_spec(spec); spec.loader.exec_module(mgf)
    man = json.load(open(os.path.join(G.GOLD, "fastq_manifest.json")))
    c, db, qs, b6, _ = G.load(man["case"])
    tmp = str(tmp_path)
    dbfa, qfq = os.path.join(tmp, "db.fa"), os.path.join(tmp, "q.fq")
    db.write_fasta(dbfa)
    mgf.write_fastq(qfq, qs)
    cli = os.path.join(os.path.dirname(capi.LIB_PATH), "ugs_cli")
    cmd = [cli, "-usearch_global", qfq, "-db", dbfa, "-id", str(c["id"]), "-strand", c["strand"], "-maxaccepts", str(c["maxaccepts"]),
           "-maxrejects", str(c["maxrejects"]), "-blast6out", os.path.join(tmp, "o.b6"), "-matchedfq", os.path.join(tmp, "o.matchedfq"),
           "-notmatchedfq", os.path.join(tmp, "o.notmatchedfq"), "-batch", "700"]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    assert open(os.path.join(tmp, "o.b6")).read() == b6
    for kind, want in man["files"].items():
        got = open(os.path.join(tmp, "o." + kind), "rb").read()
        assert got.count(b"\n") == want["lines"] and hashlib.sha256(got).hexdigest() == want["sha256"], kind
    # a FASTA query file cannot feed -matchedfq
    qfa = os.path.join(tmp, "q.fa")
    qs.write_fasta(qfa)
    r = subprocess.run([cli, "-usearch_global", qfa, "-db", dbfa, "-id", "0.9", "-strand", "plus", "-matchedfq", os.path.join(tmp, "x.fq")], stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"Cannot convert FASTA to FASTQ" in r.stderr
