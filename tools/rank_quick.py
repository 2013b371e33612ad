"""quick k_rank / k_align timing on the C2 shape for a tuning build: UGS_LIB=... python tools/rank_quick.py [queries]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usearch12_amd import capi, synth
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
AA = os.environ.get("RQ_SHAPE", "") == "aa"          # RQ_SHAPE=aa: the C5 shape (300 aa vs 2 M sequences, -id 0.8)
if AA:
    db = synth.make_db(5, 2000000, 300, aa=True)
    qs = synth.make_queries(5, db, nq, 300, aa=True)
else:
    db = synth.make_db(2, 1000000, 250)
    qs = synth.make_queries(2, db, nq, 250)
gdb = capi.UgsDB(capi.params(is_nucleo=not AA, id=0.8 if AA else 0.97), db.seqs, db.offs, device=0)
bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
bat.upload(qs.seqs, qs.offs)
r = []
for _ in range(4):
    bat.search(); bat.sync()
    st = bat.stats(); r.append((st["ms_rank"], st["ms_align"], st["ms_rank_setup"]))
h, nh, pool = bat.fetch()
import zlib
print(os.environ.get("UGS_LIB", "default"), "rank %.2f align %.2f setup %.2f" % tuple(np.min(np.array(r[1:]), axis=0)), "hits", len(h),
      "crc", zlib.crc32(h["target"].tobytes()) ^ zlib.crc32(h["ids"].tobytes()), bat.kernel_hits())
