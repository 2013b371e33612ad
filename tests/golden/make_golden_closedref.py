#!/usr/bin/env python3
"""Fixture for the closed_ref sink, produced by the UNMODIFIED reference:
oracle/_ref/usearch12 -closed_ref reads.fa -db ref.fa -strand both -tabbedout .. -threads 1
on seeded synthetic reads (size= annotations on some labels; reference families so that identity ties occur).
File: tests/golden/closedref.tab.  The reference's closed_ref sink is broken: it keeps pointers into recycled SeqInfo
objects (closedrefsink.cpp:62-63) and dies with SIGSEGV after one to two thousand queries on every input tried (and at
once with -dbotus / -dataotus).  The fixture is what it writes before that, cut at the last complete line: every line
depends only on the queries before it, so the prefix pins the per-query logic (OTU numbering, member index, top hit,
identity, ties).  Runs only where /root/reference exists."""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from usearch12_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "usearch12")
SEED, N_FAM, FAM, N_READS = 61, 60, 5, 2500


def read_label(i):
    return ("r%d;size=%d;" % (i, 1 + (i * 7) % 23)) if i % 3 else ("read%d" % i)


def ref_label(t):
    return ("ref%d;tax=f%d;" % (t, t % 11)) if t % 2 else ("R%d" % t)


def inputs():
    db, qs = synth.make_hard(SEED, N_FAM, FAM, N_READS, lmin=150, lmax=300, aa=False)
    qs = synth.revcomp_some(SEED, qs)
    return synth.SeqSet(db.seqs, db.offs, ref_label), synth.SeqSet(qs.seqs, qs.offs, read_label)


def main():
    assert os.path.exists(REF)
    db, qs = inputs()
    with tempfile.TemporaryDirectory() as tmp:
        dbfa, qfa = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.fa")
        db.write_fasta(dbfa)
        qs.write_fasta(qfa)
        rc = subprocess.call([REF, "-closed_ref", qfa, "-db", dbfa, "-strand", "both", "-tabbedout", os.path.join(HERE, "closedref.tab"),
                              "-threads", "1"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    print("reference exit code", rc)
    raw = open(os.path.join(HERE, "closedref.tab")).read()
    raw = raw[:raw.rfind("\n") + 1]                       # drop the line the crash cut short
    open(os.path.join(HERE, "closedref.tab"), "w").write(raw)
    tab = raw.splitlines()
    print(len(tab), "lines;", sum(1 for l in tab if "ties=0" not in l and not l.endswith("*")), "with ties;",
          sum(1 for l in tab if l.endswith("*")), "unassigned")


if __name__ == "__main__":
    main()
