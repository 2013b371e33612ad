"""`.udb` reader/writer.  Fixtures: tests/golden/udb_{nt,aa}.udb.gz are the reference's own
-makeudb_usearch output (tests/golden/make_golden_udb.py); udb_nt.b6/.uc the reference searching with that file as -db.

CPU part: the host-side reader parses the reference's file, and the file's letters and index rows equal the
oracle's masked DB and CSR index - which pins the ORACLE's FastMaskSeq + index build to the reference's stored bytes.
GPU part: the index built on the GPU, written by ugs_udb_write, is byte-identical to the reference's file; the CLI
searching with -db x.udb reproduces the reference's text."""
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

import golden_util
import orc
from usearch12_amd import capi, synth

CASES = json.load(open(os.path.join(golden_util.GOLD, "udb_manifest.json")))


def fixture(tmp_path, name):
    raw = gzip.open(os.path.join(golden_util.GOLD, "udb_%s.udb.gz" % name)).read()
    path = os.path.join(str(tmp_path), "ref_%s.udb" % name)
    open(path, "wb").write(raw)
    return path, raw


def inputs(name):
    c = CASES[name]
    db = synth.make_db(c["seed"], c["db_n"], c["length"], c["aa"])
    qs = synth.make_queries(c["seed"], db, c["q_n"], c["length"], c["aa"]) if c["q_n"] else None
    return c, db, qs


@pytest.mark.parametrize("name", ["nt", "aa"])
def test_reader_and_oracle_index_match_reference_file(tmp_path, name):
    path, _ = fixture(tmp_path, name)
    c, db, _ = inputs(name)
    u = capi.udb_read(path, index=True)
    assert u["is_nucleo"] == (not c["aa"]) and u["word_len"] == (5 if c["aa"] else 8)
    assert u["labels"] == db.labels()
    assert np.array_equal(u["offs"], db.offs)
    odb = orc.OrcDB(orc.params(is_nucleo=not c["aa"], id=c["id"]), db.seqs, db.offs)
    assert np.array_equal(u["seqs"], odb.masked())                     # FastMaskSeq as stored by makeudb
    ro, po = odb.index()
    assert np.array_equal(u["row_sizes"].astype(np.uint64), np.diff(ro))
    assert np.array_equal(u["postings"], po)


def test_reader_rejects_garbage(tmp_path):
    p = os.path.join(str(tmp_path), "x.udb")
    open(p, "wb").write(b"FBDU" + b"\0" * 50)
    with pytest.raises(capi.UgsError):
        capi.udb_read(p)
    with pytest.raises(capi.UgsError):
        capi.udb_read(os.path.join(str(tmp_path), "missing.udb"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["nt", "aa"])
def test_writer_byte_identical_to_reference(tmp_path, name):
    _, raw = fixture(tmp_path, name)
    c, db, _ = inputs(name)
    gdb = capi.UgsDB(capi.params(is_nucleo=not c["aa"], id=c["id"]), db.seqs, db.offs, device=0)
    out = os.path.join(str(tmp_path), "ours.udb")
    capi.udb_write(out, gdb, db.labels())
    assert open(out, "rb").read() == raw


@pytest.mark.gpu
def test_cli_udb_round_trip(tmp_path):
    path, raw = fixture(tmp_path, "nt")
    c, db, qs = inputs("nt")
    cli = os.path.join(os.path.dirname(capi.LIB_PATH), "ugs_cli")
    tmp = str(tmp_path)
    dbfa, qfa = os.path.join(tmp, "db.fa"), os.path.join(tmp, "q.fa")
    db.write_fasta(dbfa)
    qs.write_fasta(qfa)
    # -makeudb_usearch through the CLI gives the reference's file
    subprocess.check_call([cli, "-makeudb_usearch", dbfa, "-output", os.path.join(tmp, "cli.udb")])
    assert open(os.path.join(tmp, "cli.udb"), "rb").read() == raw
    # searching with the reference's .udb as -db gives the reference's text
    subprocess.check_call([cli, "-usearch_global", qfa, "-db", path, "-id", str(c["id"]), "-strand", "plus",
                           "-blast6out", os.path.join(tmp, "o.b6"), "-uc", os.path.join(tmp, "o.uc")], stderr=subprocess.DEVNULL)
    assert open(os.path.join(tmp, "o.b6")).read() == open(os.path.join(golden_util.GOLD, "udb_nt.b6")).read()
    assert open(os.path.join(tmp, "o.uc")).read() == open(os.path.join(golden_util.GOLD, "udb_nt.uc")).read()


@pytest.mark.gpu
@pytest.mark.parametrize("aa", [False, True])
def test_stored_letters_are_used_as_they_are(aa):
    """dbmask = 2 (what the CLI sets for -db x.udb): the stored, already masked letters are neither re-masked nor
    upper-cased, so the index and every search result equal the ones built from the raw FASTA letters."""
    from usearch12_amd import synth
    db, qs = synth.make_hard(77 + aa, 200, 6, 600, aa=aa)
    ident = 0.8 if aa else 0.9
    g1 = capi.UgsDB(capi.params(is_nucleo=not aa, id=ident, max_accepts=3, max_rejects=8), db.seqs, db.offs, device=0)
    masked, ro1, po1 = g1.debug_fetch()
    assert np.any((masked >= ord("a")) & (masked <= ord("z"))), "fixture without masked letters"
    g2 = capi.UgsDB(capi.params(is_nucleo=not aa, id=ident, max_accepts=3, max_rejects=8, dbmask=2), masked, db.offs, device=0)
    m2, ro2, po2 = g2.debug_fetch()
    assert np.array_equal(masked, m2) and np.array_equal(ro1, ro2) and np.array_equal(po1, po2)
    h1, n1, p1 = g1.search(qs.seqs, qs.offs)
    h2, n2, p2 = g2.search(qs.seqs, qs.offs)
    assert np.array_equal(n1, n2) and len(h1) > 300
    for f in h1.dtype.names:
        if f != "cigar_off":
            assert np.array_equal(h1[f], h2[f]), f
