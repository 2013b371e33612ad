#!/usr/bin/env python3
"""Known answers for the pair filters of Accepter::RejectPair and -abskew (-self -notself -selfid
-min_sizeratio -minqt -maxqt -minsl -maxsl -abskew), produced by the UNMODIFIED reference (oracle/_ref/usearch12
-usearch_global ... -threads 1) on seeded synthetic inputs whose labels carry ;size= annotations.  All-vs-all runs use
the database as the query file.  Both ranking paths (-big 100 forces the Big one).  Runs only where /root/reference exists."""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from usearch12_amd import synth  # noqa: E402
from make_golden import digest, REF  # noqa: E402

CASES = {
    # all-vs-all: the database is its own query set (labels equal) - the classic -self run
    "pair_self_s":   dict(seed=71, n_fam=150, fam=5, allvsall=True, id=0.9, strand="plus", opts={"self": None}, maxaccepts=2, maxrejects=8),
    "pair_self_b":   dict(seed=71, n_fam=150, fam=5, allvsall=True, id=0.9, strand="plus", opts={"self": None}, maxaccepts=2, maxrejects=8, big=100),
    "pair_selfid_b": dict(seed=72, n_fam=150, fam=5, allvsall=True, id=0.9, strand="both", opts={"selfid": None}, maxaccepts=3, maxrejects=8, big=100),
    "pair_selfid_s": dict(seed=72, n_fam=150, fam=5, allvsall=True, id=0.9, strand="both", opts={"selfid": None}, maxaccepts=3, maxrejects=8),
    "pair_len_s":    dict(seed=73, n_fam=200, fam=6, q_n=900, id=0.9, strand="plus", opts={"minqt": 0.8, "maxqt": 1.1, "minsl": 0.85}, lmin=100, lmax=400),
    "pair_len_b":    dict(seed=73, n_fam=200, fam=6, q_n=900, id=0.9, strand="both", opts={"minqt": 0.9, "maxsl": 0.99}, lmin=100, lmax=400, big=100),
    "pair_size_s":   dict(seed=74, n_fam=200, fam=6, q_n=900, id=0.9, strand="plus", opts={"min_sizeratio": 2.0}, maxaccepts=2, maxrejects=16),
    "pair_size_b":   dict(seed=74, n_fam=200, fam=6, q_n=900, id=0.9, strand="plus", opts={"min_sizeratio": 0.5, "abskew": 1.5}, maxaccepts=2, maxrejects=16, big=100),
    "pair_notself_b": dict(seed=75, n_fam=100, fam=4, allvsall=True, id=0.9, strand="plus", opts={"notself": None}, big=100),
}


def size_of(i, salt, always=False):
    """;size= annotation of sequence i; some labels have none unless `always` (the reference dies on a missing size when
    -min_sizeratio / -abskew need it, label.cpp:152-161)"""
    return None if ((i + salt) % 5 == 0 and not always) else 1 + (i * 7 + salt * 3) % 40


def make_inputs(c):
    always = any(k in c["opts"] for k in ("min_sizeratio", "abskew"))
    db, qs = synth.make_hard(c["seed"], c["n_fam"], c["fam"], c.get("q_n", 10), lmin=c.get("lmin", 150), lmax=c.get("lmax", 400), aa=False)

    def tl(i):
        s = size_of(i, 1, always)
        return "t%d" % i if s is None else "t%d;size=%d;" % (i, s)

    def ql(i):
        s = size_of(i, 2, always)
        return "q%d" % i if s is None else "q%d;size=%d;" % (i, s)
    db = synth.SeqSet(db.seqs, db.offs, tl)
    if c.get("allvsall"):
        return db, db
    if c.get("strand") == "both":
        qs = synth.revcomp_some(c["seed"], qs)
    return db, synth.SeqSet(qs.seqs, qs.offs, ql)


def ref_cmd(c, qfa, dbfa, prefix):
    cmd = [REF, "-usearch_global", qfa, "-db", dbfa, "-id", str(c["id"]), "-blast6out", prefix + ".b6", "-threads", "1", "-strand", c["strand"]]
    for opt in ("big", "maxaccepts", "maxrejects"):
        if opt in c:
            cmd += ["-" + opt, str(c[opt])]
    for k, v in c["opts"].items():
        cmd += ["-" + k] + ([] if v is None else [str(v)])
    return cmd


def main():
    assert os.path.exists(REF)
    manifest = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, c in CASES.items():
            db, qs = make_inputs(c)
            dbfa, qfa = os.path.join(tmp, "db.fa"), os.path.join(tmp, "q.fa")
            db.write_fasta(dbfa)
            qs.write_fasta(qfa)
            prefix = os.path.join(HERE, name)
            cmd = ref_cmd(c, qfa, dbfa, prefix)
            subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            n = sum(1 for _ in open(prefix + ".b6"))
            manifest[name] = dict(c, db_sha256=digest(db), q_sha256=digest(qs), n_hits=n,
                                  cmd=" ".join(["usearch12"] + [os.path.basename(x) if x.startswith(tmp) else x for x in cmd[1:]]).replace(HERE + "/", ""))
            print(name, "hits", n)
    json.dump(manifest, open(os.path.join(HERE, "pairs_manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
