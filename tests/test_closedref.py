"""closed_ref sink.  Expected text: the reference's own -tabbedout (tests/golden/make_golden_closedref.py) -
the prefix it writes before its sink crashes (see the generator), 1 984 of 2 500 queries.
CPU part: libugs.so's sink fed with the ORACLE's hits (cmd_closed_ref defaults: -id 0.97 -stepwords 0, terminator 4/16).
GPU part: `ugs_cli -closed_ref` end to end."""
import ctypes as C
import importlib.util
import os
import subprocess

import pytest

import golden_util as G
import orc
from usearch12_amd import capi

_spec = importlib.util.spec_from_file_location("mgc", os.path.join(G.GOLD, "make_golden_closedref.py"))
mgc = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mgc)
GOLD = open(os.path.join(G.GOLD, "closedref.tab"), "rb").read()


def test_closedref_from_oracle_hits():
    L = capi.lib()
    L.ugs_closedref_create.restype = C.c_void_p
    L.ugs_closedref_add.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_char_p), C.c_char_p, C.c_int]
    L.ugs_closedref_destroy.argtypes = [C.c_void_p]
    L.ugs_closedref_totals.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    db, qs = mgc.inputs()
    p = orc.params(is_nucleo=True, id=0.97, strand_both=1, max_accepts=4, max_rejects=16, stepwords=0)
    hits, nh, _ = orc.OrcDB(p, db.seqs, db.offs).search(qs.seqs, qs.offs, nthreads=4)
    ql, tl = qs.labels(), db.labels()
    sink = L.ugs_closedref_create()
    buf = C.create_string_buffer(1 << 14)
    lines, k = [], 0
    for qi in range(qs.n):
        n = int(nh[qi])
        h = hits[k:k + n]
        k += n
        labs = (C.c_char_p * max(n, 1))(*[tl[int(t)].encode() for t in h["target"]])
        ln = L.ugs_closedref_add(sink, ql[qi].encode(), h.ctypes.data if n else None, n, labs, buf, len(buf))
        assert ln > 0
        lines.append(buf.raw[:ln])
    a, u, o = C.c_uint64(), C.c_uint64(), C.c_uint32()
    L.ugs_closedref_totals(sink, C.byref(a), C.byref(u), C.byref(o))
    L.ugs_closedref_destroy(sink)
    got = b"".join(lines)
    assert len(GOLD) > 50000 and got[:len(GOLD)] == GOLD
    assert a.value + u.value == qs.n and 0 < o.value <= db.n
    assert sum(1 for x in lines if b"ties=0" not in x and not x.endswith(b"*\n")) >= 10      # ties are exercised


@pytest.mark.gpu
def test_cli_closed_ref_identical_to_reference(tmp_path):
    db, qs = mgc.inputs()
    tmp = str(tmp_path)
    dbfa, qfa = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.fa")
    db.write_fasta(dbfa)
    qs.write_fasta(qfa)
    cli = os.path.join(os.path.dirname(capi.LIB_PATH), "ugs_cli")
    subprocess.check_call([cli, "-closed_ref", qfa, "-db", dbfa, "-strand", "both", "-tabbedout", os.path.join(tmp, "o.tab")], stderr=subprocess.DEVNULL)
    got = open(os.path.join(tmp, "o.tab"), "rb").read()
    assert got[:len(GOLD)] == GOLD and got.count(b"\n") == qs.n
