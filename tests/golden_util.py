"""Load golden cases (tests/golden/manifest.json): regenerate the seeded inputs, check their
digests, return inputs + the reference's expected blast6/uc text."""
import hashlib
import importlib.util
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))

_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
_mg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mg)


def case_names(pred=None):
    return [n for n, c in sorted(MANIFEST.items()) if pred is None or pred(c)]


def is_big(c):
    """True when the reference takes the Big ranking path for this case."""
    n = c["db_n"] if c["gen"] == "uniform" else c["n_fam"] * c["fam"]
    return n > c.get("big", 100000)


def load(name):
    c = MANIFEST[name]
    db, qs = _mg.make_inputs(c)
    assert _mg.digest(db) == c["db_sha256"], "generator drift (db) for " + name
    assert _mg.digest(qs) == c["q_sha256"], "generator drift (queries) for " + name
    b6 = open(os.path.join(GOLD, name + ".b6")).read()
    uc = open(os.path.join(GOLD, name + ".uc")).read()
    return c, db, qs, b6, uc


def params_kw(c):
    kw = dict(strand_both=1 if c.get("strand") == "both" else 0)
    if "big" in c:
        kw["big"] = c["big"]
    if "maxaccepts" in c:
        kw["max_accepts"] = c["maxaccepts"]
    if "maxrejects" in c:
        kw["max_rejects"] = c["maxrejects"]
    if "band" in c:
        kw["band"] = c["band"]
    if c.get("fulldp") or c.get("gaforce"):
        kw["align_flags"] = (1 if c.get("fulldp") else 0) | (2 if c.get("gaforce") else 0)
    if c.get("hardmask"):
        kw["dbmask"] = 3
    for opt, bit in (("termid", 4), ("termidd", 8)):
        if opt in c:
            kw["align_flags"] = kw.get("align_flags", 0) | bit
            kw[opt] = c[opt]
    for opt, field in (("wordlength", "word_len"), ("stepwords", "stepwords"), ("bump", "bump_pct"), ("minhsp", "minhsp"), ("xdrop_nw", "xdrop_nw"),
                       ("match", "match"), ("mismatch", "mismatch"), ("hspw", "hsp_word_len")):
        if opt in c:
            kw[field] = c[opt]
    if "dbmask" in c:                   # -dbmask none = everything upper case (0), user = letters as given (2)
        kw["dbmask"] = {"none": 0, "user": 2}[c["dbmask"]]
    for opt in _mg.FILTER_OPTS:         # optional accept filters (params() sets the filter_mask bit)
        if opt in c:
            kw[opt] = c[opt]
    return kw


def load_xdrop(name):
    """tests/golden/xdrop_{nt,aa}.txt -> list of dict(mode, x, a, b, anc, want) where want is the
    reference's answer: (score, loi, loj, leni, lenj, path) with loi/loj None for F/B lines."""
    out = []
    lines = open(os.path.join(GOLD, "xdrop_%s.txt" % name)).read().splitlines()
    for k in range(0, len(lines), 2):
        c = lines[k].split()
        r = lines[k + 1].split()
        assert r[0] == "="
        mode, x, a, b = c[0], float(c[1]), c[2], c[3]
        anc = tuple(int(v) for v in c[4:7]) if mode == "A" else (0, 0, 0)
        if mode == "A":
            want = (float(r[1]), int(r[2]), int(r[3]), int(r[4]), int(r[5]), "" if r[6] == "-" else r[6])
        else:
            want = (float(r[1]), None, None, int(r[2]), int(r[3]), "" if r[4] == "-" else r[4])
        out.append(dict(mode=mode, x=x, a=a, b=b, anc=anc, want=want))
    return out


# ---- usearch_local cases (tests/golden/local_manifest.json, make_golden_local.py)
LOCAL_MANIFEST = json.load(open(os.path.join(GOLD, "local_manifest.json")))
_spec_l = importlib.util.spec_from_file_location("make_golden_local", os.path.join(GOLD, "make_golden_local.py"))
_mgl = importlib.util.module_from_spec(_spec_l)
_spec_l.loader.exec_module(_mgl)


def local_case_names():
    return sorted(LOCAL_MANIFEST)


def load_local(name):
    c = LOCAL_MANIFEST[name]
    db, qs = _mgl.make_inputs(c)
    assert _mg.digest(db) == c["db_sha256"], "generator drift (db) for " + name
    assert _mg.digest(qs) == c["q_sha256"], "generator drift (queries) for " + name
    return c, db, qs, open(os.path.join(GOLD, name + ".b6")).read()


def local_params_kw(c):
    """keyword arguments for orc.params()/capi.params(): usearch_local with or without -id"""
    kw = params_kw(c)
    kw["local_evalue"] = c["evalue"]
    kw["id"] = c.get("id")
    for opt, field, sign in (("xdrop_u", "xdrop_u", 1), ("xdrop_g", "xdrop_g", 1), ("maxhsps", "max_hsps", 1), ("ka_dbsize", "ka_dbsize", 1), ("hspw", "hsp_word_len", 1),
                             ("lopen", "local_open", -1), ("lext", "local_ext", -1), ("match", "match", 1), ("mismatch", "mismatch", 1)):
        if opt in c:
            kw[field] = sign * c[opt] if sign < 0 else c[opt]
    return kw


# ---- pair-filter cases (tests/golden/pairs_manifest.json, make_golden_pairs.py)
PAIRS_MANIFEST = json.load(open(os.path.join(GOLD, "pairs_manifest.json")))
_spec_p = importlib.util.spec_from_file_location("make_golden_pairs", os.path.join(GOLD, "make_golden_pairs.py"))
_mgp = importlib.util.module_from_spec(_spec_p)
_spec_p.loader.exec_module(_mgp)


def pair_case_names():
    return sorted(PAIRS_MANIFEST)


def load_pairs(name):
    c = PAIRS_MANIFEST[name]
    db, qs = _mgp.make_inputs(c)
    assert _mg.digest(db) == c["db_sha256"] and _mg.digest(qs) == c["q_sha256"], "generator drift for " + name
    return c, db, qs, open(os.path.join(GOLD, name + ".b6")).read()


def pair_params_kw(c):
    kw = params_kw(c)
    for k, v in c["opts"].items():
        kw[k] = True if v is None else v
    return kw


def pair_keys(db, qs):
    """(db label keys, db sizes, query label keys, query sizes): interned labels and ;size= annotations (UINT32_MAX = none)"""
    ids = {}

    def keys(ss):
        k = np.zeros(ss.n, np.uint32)
        z = np.full(ss.n, 0xffffffff, np.uint32)
        for i, lab in enumerate(ss.labels()):
            k[i] = ids.setdefault(lab, len(ids))
            j = lab.find(";size=")
            if j >= 0:
                z[i] = int(lab[j + 6:].split(";")[0])
        return k, z
    tk, tz = keys(db)
    qk, qz = keys(qs)
    return tk, tz, qk, qz


# ---- cluster_fast cases (tests/golden/cluster_manifest.json, make_golden_cluster.py)
def cluster_cases():
    return sorted(json.load(open(os.path.join(GOLD, "cluster_manifest.json"))).keys())


def load_cluster(name):
    """-> (case dict, reads SeqSet, the reference's -uc text, its -centroids text)"""
    import gzip
    spec = importlib.util.spec_from_file_location("make_golden_cluster", os.path.join(GOLD, "make_golden_cluster.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    c = json.load(open(os.path.join(GOLD, "cluster_manifest.json")))[name]
    r = mg.make_reads(c)
    assert mg.digest(r) == c["reads_sha256"], "generator drift (reads) for " + name
    uc = gzip.open(os.path.join(GOLD, name + ".uc.gz"), "rb").read().decode()
    cen = gzip.open(os.path.join(GOLD, name + ".cent.fa.gz"), "rb").read().decode()
    return c, r, uc, cen
