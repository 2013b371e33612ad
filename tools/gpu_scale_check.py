"""Medium-size GPU-vs-oracle checks on shapes the golden cases do not reach (run on the GPU box)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
from usearch12_amd import capi, synth


def check(name, db, qs, aa, ident, **kw):
    t0 = time.time()
    p = capi.params(is_nucleo=not aa, id=ident, **kw)
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
    bat.upload(qs.seqs, qs.offs); bat.search(); bat.sync()
    hits, nh, pool = bat.fetch()
    st = bat.stats()
    t1 = time.time()
    odb = orc.OrcDB(orc.params(is_nucleo=not aa, id=ident, **kw), db.seqs, db.offs)
    oh, onh, opool = odb.search(qs.seqs, qs.offs, nthreads=min(64, os.cpu_count() or 1))
    t2 = time.time()
    ok = np.array_equal(nh, onh) and all(np.array_equal(hits[f], oh[f]) for f in hits.dtype.names if f != "cigar_off")
    if ok:
        for a, b in zip(hits[::101], oh[::101]):
            ok = ok and np.array_equal(pool[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["cigar_len"])],
                                       opool[int(b["cigar_off"]):int(b["cigar_off"]) + int(b["cigar_len"])])
    print("%-14s %s hits %d/%d  gpu %.2fs (rank %.1f ms, align %.1f ms)  oracle %.1fs" %
          (name, "OK" if ok else "MISMATCH", len(hits), len(oh), t1 - t0, st["ms_rank"], st["ms_align"], t2 - t1), flush=True)
    return ok


ok = True
db = synth.make_db(5, 200_000, 300, aa=True); qs = synth.make_queries(5, db, 50_000, 300, aa=True)
ok &= check("aa_200k", db, qs, True, 0.8)
db = synth.make_db(31, 150_000, 250); qs = synth.revcomp_some(31, synth.make_queries(31, db, 50_000, 250))
ok &= check("nt_both_150k", db, qs, False, 0.97, strand_both=1)
db, qs = synth.make_hard(32, 2000, 10, 4000, lmin=1200, lmax=1600)
ok &= check("nt_long_1500", db, qs, False, 0.97, big=100)
db, qs = synth.make_hard(33, 20000, 8, 20000, lmin=150, lmax=400)
ok &= check("hard_160k", db, qs, False, 0.95)
db = synth.make_db(34, 60_000, 250); qs = synth.make_queries(34, db, 20_000, 250)
ok &= check("nt_small_60k", db, qs, False, 0.97)
print("ALL OK" if ok else "FAILED")
