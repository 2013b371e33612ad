import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_fuzz as fz
from usearch12_amd import capi, synth
seed = int(sys.argv[1])
aa, lmin, lmax, n_fam, fam, nq, ident, kw = fz.config(seed)
print(seed, aa, lmin, lmax, n_fam, fam, nq, ident, kw, flush=True)
db, qs = synth.make_hard(1000 + seed, n_fam, fam, nq, lmin=lmin, lmax=lmax, aa=aa)
if kw.get("strand_both"):
    qs = synth.revcomp_some(seed, qs)
print("db", db.n, "q", qs.n, "maxq", int(np.diff(qs.offs.astype(np.int64)).max()), "minq", int(np.diff(qs.offs.astype(np.int64)).min()), flush=True)
gdb = capi.UgsDB(capi.params(is_nucleo=not aa, id=ident, **kw), db.seqs, db.offs, device=0)
m, ro, po = gdb.debug_fetch()
print("db ok, postings", len(po), flush=True)
hits, nh, pool = gdb.search(qs.seqs, qs.offs)
print("hits", len(hits))
