"""otutab sink.  Expected files: the reference's own -otutabout / -mapout (tests/golden/make_golden_otutab.py).
CPU part: libugs.so's OTU-table functions fed with the ORACLE's hits (cmd_otutab defaults: -id 0.97 -maxaccepts 3
-maxrejects 32 -stepwords 0 -strand both) reproduce both files.  GPU part: `ugs_cli -otutab` end to end."""
import ctypes as C
import importlib.util
import os
import subprocess

import numpy as np
import pytest

import golden_util as G
import orc
from usearch12_amd import capi

_spec = importlib.util.spec_from_file_location("mgo", os.path.join(G.GOLD, "make_golden_otutab.py"))
mgo = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mgo)


def gold(name):
    return open(os.path.join(G.GOLD, name), "rb").read()


def test_otutab_from_oracle_hits(tmp_path):
    L = capi.lib()
    L.ugs_otutab_create.restype = C.c_void_p
    L.ugs_otutab_add.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
    L.ugs_otutab_write.argtypes = [C.c_void_p, C.c_char_p]
    L.ugs_otutab_destroy.argtypes = [C.c_void_p]
    L.ugs_hits_to_report.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    L.ugs_hits_to_report.restype = C.c_uint32
    db, qs = mgo.inputs()
    p = orc.params(is_nucleo=True, id=0.97, strand_both=1, max_accepts=3, max_rejects=32, stepwords=0)
    hits, nh, _ = orc.OrcDB(p, db.seqs, db.offs).search(qs.seqs, qs.offs, nthreads=4)
    ql, tl = qs.labels(), db.labels()
    tab = L.ugs_otutab_create()
    buf = C.create_string_buffer(1 << 12)
    lines, k = [], 0
    for qi in range(qs.n):
        n = int(nh[qi])
        h = hits[k:k + n]
        k += n
        top = C.c_uint32(0)
        if n:
            L.ugs_hits_to_report(h.ctypes.data, n, 0, 1, 0, C.byref(top))
        ln = L.ugs_otutab_add(tab, ql[qi].encode(), tl[int(h[top.value]["target"])].encode() if n else None, buf, len(buf))
        assert ln >= 0
        lines.append(buf.raw[:ln])
    out = os.path.join(str(tmp_path), "t.tab")
    assert L.ugs_otutab_write(tab, out.encode()) == 0
    L.ugs_otutab_write_biom.argtypes = [C.c_void_p, C.c_char_p]
    biom = os.path.join(str(tmp_path), "t.biom")
    assert L.ugs_otutab_write_biom(tab, biom.encode()) == 0
    L.ugs_otutab_destroy(tab)
    # -biomout (json.cpp:32-103); the "id" (= output path) and "date" lines differ from run to run
    got = b"".join(ln for ln in open(biom, "rb").read().splitlines(True) if not ln.startswith((b'\t"id"', b'\t"date"')))
    assert got == gold("otutab.biom")
    assert b"".join(lines) == gold("otutab.map")
    assert open(out, "rb").read() == gold("otutab.tab")


@pytest.mark.gpu
def test_cli_otutab_identical_to_reference(tmp_path):
    db, qs = mgo.inputs()
    tmp = str(tmp_path)
    dbfa, qfa = os.path.join(tmp, "otus.fa"), os.path.join(tmp, "reads.fa")
    db.write_fasta(dbfa)
    qs.write_fasta(qfa)
    cli = os.path.join(os.path.dirname(capi.LIB_PATH), "ugs_cli")
    subprocess.check_call([cli, "-otutab", qfa, "-otus", dbfa, "-otutabout", os.path.join(tmp, "o.tab"), "-mapout", os.path.join(tmp, "o.map"),
                           "-biomout", os.path.join(tmp, "o.biom")], stderr=subprocess.DEVNULL)
    got = b"".join(ln for ln in open(os.path.join(tmp, "o.biom"), "rb").read().splitlines(True) if not ln.startswith((b'\t"id"', b'\t"date"')))
    assert got == gold("otutab.biom")
    assert open(os.path.join(tmp, "o.map"), "rb").read() == gold("otutab.map")
    assert open(os.path.join(tmp, "o.tab"), "rb").read() == gold("otutab.tab")
