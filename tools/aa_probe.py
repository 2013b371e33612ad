import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usearch12_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
db = synth.make_db(5, n, 300, aa=True); qs = synth.make_queries(5, db, 50_000, 300, aa=True)
gdb = capi.UgsDB(capi.params(is_nucleo=False, id=0.8), db.seqs, db.offs, device=0)
bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
bat.upload(qs.seqs, qs.offs)
for _ in range(2):
    bat.search(); bat.sync(); st = bat.stats()
print(os.environ.get("UGS_GSIZE"), {k: round(st[k], 2) for k in ("ms_rank", "ms_rank_setup", "ms_align")})
