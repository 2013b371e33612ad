#!/usr/bin/env python3
"""FASTQ queries through the UNMODIFIED reference (oracle/_ref/usearch12 -usearch_global reads.fq ... -threads 1): the golden
case hard_acc_s with seeded quality strings; -blast6out, -matchedfq and -notmatchedfq are kept (sha256 + first lines).
Runs only where /root/reference exists."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

CASE = "hard_acc_s"


def write_fastq(path, qs, seed=5):
    rng = np.random.default_rng(seed)
    with open(path, "wb") as f:
        for i in range(qs.n):
            s = qs.seq(i)
            q = bytes(rng.integers(35, 74, len(s)).astype(np.uint8))
            f.write(b"@" + qs.label(i).encode() + b"\n" + s + b"\n+\n" + q + b"\n")


def main():
    assert os.path.exists(mg.REF)
    c = mg.CASES[CASE]
    db, qs = mg.make_inputs(c)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        dbfa, qfq = os.path.join(tmp, "db.fa"), os.path.join(tmp, "q.fq")
        db.write_fasta(dbfa)
        write_fastq(qfq, qs)
        cmd = [mg.REF, "-usearch_global", qfq, "-db", dbfa, "-id", str(c["id"]), "-strand", c["strand"], "-threads", "1",
               "-maxaccepts", str(c["maxaccepts"]), "-maxrejects", str(c["maxrejects"])]
        for kind, opt in (("b6", "-blast6out"), ("matchedfq", "-matchedfq"), ("notmatchedfq", "-notmatchedfq")):
            cmd += [opt, os.path.join(tmp, "o." + kind)]
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for kind in ("b6", "matchedfq", "notmatchedfq"):
            data = open(os.path.join(tmp, "o." + kind), "rb").read()
            out[kind] = dict(sha256=hashlib.sha256(data).hexdigest(), lines=data.count(b"\n"))
            if kind != "b6":
                open(os.path.join(HERE, "fastq_" + kind + ".head"), "wb").write(b"".join(data.splitlines(True)[:8]))
    json.dump(dict(case=CASE, files=out), open(os.path.join(HERE, "fastq_manifest.json"), "w"), indent=1, sort_keys=True)
    print(out)


if __name__ == "__main__":
    main()
