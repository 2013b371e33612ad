#!/usr/bin/env python3
"""Golden vectors for cluster_fast: the compiled, UNMODIFIED reference (oracle/_ref/usearch12) run with
-threads 1 on seeded synthetic reads.  Committed per case: <case>.uc.gz and <case>.cent.fa.gz (the reference's -uc and
-centroids files, gzip -n) and cluster_manifest.json with the generator arguments, the command line and the sha256 of the
generated reads.  Inputs are regenerated from usearch12_amd/synth.py by the tests."""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from usearch12_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "usearch12")

CASES = {
    # small ranking path only (never more than -big centroids)
    "cl_small":  dict(gen="reads", seed=41, n=4000, species=40, dup=0.05, id=0.97, strand="plus"),
    # crosses the small -> Big latch (udbusortedsearcher.cpp:39-58) with a lowered -big
    "cl_latch":  dict(gen="reads", seed=42, n=6000, species=60, dup=0.03, id=0.97, strand="plus", big=300),
    # both strands: half of the reads reverse-complemented, dereplication and search see both orientations
    "cl_both":   dict(gen="reads", seed=43, n=5000, species=50, dup=0.05, id=0.97, strand="both", big=250),
    # lower identity, noisier reads: deeper candidate walks, more rejects before an accept
    "cl_id90":   dict(gen="reads", seed=44, n=5000, species=40, dup=0.02, id=0.90, strand="plus", big=200, sub=0.03, indel=0.004),
    # variable lengths, lower-case stretches and ambiguity codes (words voided, IUPAC identities), both strands
    "cl_hard":   dict(gen="hard", seed=45, n=4000, id=0.95, strand="both", big=300),
    # one dominant species: long chains of centroids founded inside one batch of the device loop
    "cl_skew":   dict(gen="reads", seed=46, n=6000, species=3, dup=0.0, id=0.97, strand="plus", big=400),
    # -sort length (GetSeqOrder clusterfast.cpp:37-79): variable read lengths with many ties, QuickSortOrderDesc's tie order
    "cl_sortlen": dict(gen="hard", seed=47, n=3000, id=0.95, strand="plus", big=300, sort="length"),
    "cl_sortlen2": dict(gen="reads", seed=48, n=5000, species=40, dup=0.05, id=0.97, strand="both", big=200, sort="length", indel=0.004),
    # ;size= annotations on two thirds of the labels: -sort size reads them with default 1, -sizeout strips and re-appends
    "cl_sortsize": dict(gen="reads", seed=49, n=5000, species=40, dup=0.05, id=0.97, strand="plus", big=250, sort="size", sizes=0.66, sizeout=1),
    # every label annotated: -sizein sums them into the cluster sizes (C records, centroid order), -sizeout, -minsize
    "cl_sizein": dict(gen="reads", seed=50, n=5000, species=40, dup=0.05, id=0.97, strand="plus", big=250, sort="size", sizes=1.0, sizein=1,
                      sizeout=1, minsize=3),
    "cl_sizein_nosort": dict(gen="reads", seed=51, n=3000, species=30, dup=0.05, id=0.97, strand="both", sizes=1.0, sizein=1),
    # protein (UCLUST's other everyday use): families of 20-400 residue sequences, small path only / across the small -> Big latch
    # (sparse index: on the device the Big-phase searches are ranked by k_rank), a lower identity with deeper walks
    "cl_aa_small": dict(gen="hard", seed=52, n=3000, id=0.9, strand="plus", aa=1),
    "cl_aa_latch": dict(gen="hard", seed=53, n=4000, id=0.9, strand="plus", aa=1, big=300),
    "cl_aa_id70":  dict(gen="hard", seed=54, n=3000, id=0.7, strand="plus", aa=1, big=250),
    # -maxrejects other than the command's 8 (terminator.cpp:22-31): deeper walks before a read founds a cluster, up to the 64 candidates
    # of a ranking pass; and a walk that gives up after two rejects
    "cl_rej32":    dict(gen="reads", seed=55, n=5000, species=40, dup=0.02, id=0.90, strand="both", big=200, sub=0.04, indel=0.005, maxrejects=32),
    "cl_rej64_aa": dict(gen="hard", seed=56, n=3000, id=0.8, strand="plus", aa=1, big=250, maxrejects=64),
    "cl_rej2":     dict(gen="reads", seed=57, n=4000, species=30, dup=0.02, id=0.95, strand="plus", big=200, sub=0.03, maxrejects=2),
}


def with_sizes(r, seed, frac):
    """append ;size=N; (or ;size=N without the closing separator, or in the middle of other annotations) to a fraction of the labels"""
    import numpy as np
    rng = np.random.default_rng(seed + 1000)
    pick = rng.random(r.n) < frac
    val = np.where(rng.random(r.n) < 0.6, rng.integers(1, 4, r.n), rng.integers(1, 400, r.n))
    style = rng.integers(0, 3, r.n)
    base = r._label_fn

    def lab(i):
        l = base(i)
        if not pick[i]:
            return l
        if style[i] == 0:
            return "%s;size=%d;" % (l, val[i])
        if style[i] == 1:
            return "%s;size=%d" % (l, val[i])
        return "%s;sample=s%d;size=%d;tax=x;" % (l, i % 7, val[i])
    return synth.SeqSet(r.seqs, r.offs, lab)


def make_reads(c):
    if c["gen"] == "reads":
        r = synth.make_reads(c["seed"], c["n"], n_species=c["species"], dup_frac=c["dup"],
                             p_sub=c.get("sub", 0.01), p_del=c.get("indel", 0.001), p_ins=c.get("indel", 0.001))
    else:
        _, r = synth.make_hard(c["seed"], 200, 6, c["n"], lmin=120, lmax=400, aa=bool(c.get("aa")))
    if c["strand"] == "both":
        r = synth.revcomp_some(c["seed"], r)
    if c.get("sizes"):
        r = with_sizes(r, c["seed"], c["sizes"])
    return r


def digest(ss):
    h = hashlib.sha256()
    h.update(ss.offs.tobytes())
    h.update(ss.seqs.tobytes())
    if getattr(ss, "_label_fn", None) is not None and ss.n and ";size=" in "".join(ss.label(i) for i in range(min(ss.n, 64))):
        h.update("\n".join(ss.labels()).encode())
    return h.hexdigest()


def gz_write(path, data):
    with open(path, "wb") as f:
        with gzip.GzipFile(fileobj=f, mode="wb", mtime=0, filename="") as g:
            g.write(data)


def main():
    assert os.path.exists(REF), "build the reference first: oracle/build_ref.sh"
    only = set(sys.argv[1:])
    manifest = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, c in CASES.items():
            if only and name not in only:
                continue
            r = make_reads(c)
            fa = os.path.join(tmp, "reads.fa")
            r.write_fasta(fa)
            uc, cen = os.path.join(tmp, "o.uc"), os.path.join(tmp, "o.cent.fa")
            cmd = [REF, "-cluster_fast", fa, "-id", str(c["id"]), "-uc", uc, "-centroids", cen, "-threads", "1", "-strand", c["strand"]]
            if "big" in c:
                cmd += ["-big", str(c["big"])]
            if c.get("sort"):
                cmd += ["-sort", c["sort"]]
            for flag in ("sizein", "sizeout"):
                if c.get(flag):
                    cmd += ["-" + flag]
            if c.get("minsize"):
                cmd += ["-minsize", str(c["minsize"])]
            if c.get("maxrejects"):
                cmd += ["-maxrejects", str(c["maxrejects"])]
            subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            uct = open(uc, "rb").read()
            gz_write(os.path.join(HERE, name + ".uc.gz"), uct)
            gz_write(os.path.join(HERE, name + ".cent.fa.gz"), open(cen, "rb").read())
            ncl = sum(1 for ln in uct.split(b"\n") if ln.startswith(b"C\t"))
            manifest[name] = dict(c, reads_sha256=digest(r), n_clusters=ncl,
                                  cmd=" ".join(["usearch12"] + [os.path.basename(x) if x.startswith(tmp) else x for x in cmd[1:]]))
            print(name, "clusters", ncl)
    path = os.path.join(HERE, "cluster_manifest.json")
    if only and os.path.exists(path):
        old = json.load(open(path))
        old.update(manifest)
        manifest = old
    json.dump(manifest, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
