"""k_align on the C2 shape with queries that are all random / all have a relative: UGS_LIB=... python tools/align_rand.py [queries]  (phase clocks with a -DUGS_ALIGN_CLOCKS build)"""
import os, sys, zlib, numpy as np
sys.path.insert(0, "/root/repo")
from usearch12_amd import capi, synth
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
db = synth.make_db(2, 1000000, 250)
gdb = capi.UgsDB(capi.params(is_nucleo=True, id=0.97), db.seqs, db.offs, device=0)
for fr in (1.0, 0.0):
    qs = synth.make_queries(2, db, nq, 250, frac_random=fr)
    bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
    bat.upload(qs.seqs, qs.offs)
    r = []
    for _ in range(3):
        bat.search(); bat.sync()
        st = bat.stats(); r.append(st["ms_align"])
    h, nh, pool = bat.fetch()
    print(os.environ.get("UGS_LIB", "default"), "group", os.environ.get("UGS_ALIGN_GROUP", "-"), "frac_random", fr, "align %.3f ms" % min(r[1:]), "pairs", st["pairs_aligned"], "ns/pair %.3f" % (min(r[1:]) * 1e6 / max(1, st["pairs_aligned"])),
          "hits", len(h), "crc", zlib.crc32(h["target"].tobytes()) ^ zlib.crc32(h["ids"].tobytes()), flush=True)
