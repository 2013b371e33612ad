#!/usr/bin/env python3
"""Fixture for the otutab sink ('otutab/closed_ref sinks reuse the global path unchanged'), produced by
the UNMODIFIED reference: oracle/_ref/usearch12 -otutab reads.fa -otus otus.fa -otutabout .. -mapout .. -threads 1 on
seeded synthetic reads whose labels carry sample names (three spellings: sample=, barcodelabel=, leading word) and
size= annotations.  Files: tests/golden/otutab.tab / otutab.map.  Runs only where /root/reference exists."""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from usearch12_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "usearch12")
SEED, N_FAM, FAM, N_READS = 51, 120, 4, 3000


def read_label(i):
    s = i % 7
    if i % 3 == 0:
        return "r%d;sample=S%d;size=%d;" % (i, s, 1 + i % 5)
    if i % 3 == 1:
        return "r%d;barcodelabel=B%d;" % (i, s)
    return "Smp%d.%d;size=%d;" % (s, i, 2 + i % 3)


def otu_label(t):
    return ("Otu%d;size=%d;" % (t + 1, 100 + t)) if t % 4 else ("Z%d;otu=Zotu%d;" % (t, t + 1))


def inputs():
    db, qs = synth.make_hard(SEED, N_FAM, FAM, N_READS, lmin=150, lmax=400, aa=False)
    qs = synth.revcomp_some(SEED, qs)
    db = synth.SeqSet(db.seqs, db.offs, otu_label)
    qs = synth.SeqSet(qs.seqs, qs.offs, read_label)
    return db, qs


def main():
    assert os.path.exists(REF)
    db, qs = inputs()
    with tempfile.TemporaryDirectory() as tmp:
        dbfa, qfa = os.path.join(tmp, "otus.fa"), os.path.join(tmp, "reads.fa")
        db.write_fasta(dbfa)
        qs.write_fasta(qfa)
        subprocess.check_call([REF, "-otutab", qfa, "-otus", dbfa, "-otutabout", os.path.join(HERE, "otutab.tab"), "-mapout",
                               os.path.join(HERE, "otutab.map"), "-biomout", os.path.join(tmp, "o.biom"), "-threads", "1"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        # -biomout: the "id" (output path) and "date" lines vary from run to run and are left out of the fixture
        biom = [ln for ln in open(os.path.join(tmp, "o.biom")).read().splitlines(True) if not ln.startswith(('\t"id"', '\t"date"'))]
        open(os.path.join(HERE, "otutab.biom"), "w").write("".join(biom))
    print(os.path.getsize(os.path.join(HERE, "otutab.tab")), os.path.getsize(os.path.join(HERE, "otutab.map")))


if __name__ == "__main__":
    main()
