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
