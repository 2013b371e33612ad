"""End-to-end wall time of the C++ driver (FASTA in, blast6 + uc out) on the C2 shape, for the record."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usearch12_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
db = synth.make_db(2, n, 250); qs = synth.make_queries(2, db, n, 250)
with tempfile.TemporaryDirectory() as tmp:
    dbfa, qfa = os.path.join(tmp, "db.fa"), os.path.join(tmp, "q.fa")
    db.write_fasta(dbfa); qs.write_fasta(qfa)
    cli = os.path.join(ROOT, "usearch12_amd", "ugs_cli")
    t0 = time.time()
    subprocess.check_call([cli, "-usearch_global", qfa, "-db", dbfa, "-id", "0.97", "-strand", "plus", "-blast6out", os.path.join(tmp, "o.b6"), "-uc", os.path.join(tmp, "o.uc")])
    t1 = time.time()
    print("ugs_cli end to end: %.2f s for %d queries vs %d targets (%.0f q/s incl. FASTA parsing, index build, text output)" % (t1 - t0, n, n, n / (t1 - t0)))
