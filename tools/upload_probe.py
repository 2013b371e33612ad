"""how long does one batch upload take (H2D of the query letters + offsets), alone and beside a running search"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from usearch12_amd import capi, synth
db = synth.make_db(2, 1000000, 250)
qs = synth.make_queries(2, db, 1000000, 250)
gdb = capi.UgsDB(capi.params(is_nucleo=True, id=0.97), db.seqs, db.offs, device=0)
b0 = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1])); b1 = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
for reg in (False, True):
    if reg:
        capi._chk(capi.lib().ugs_host_register(qs.seqs.ctypes.data, qs.seqs.nbytes))
    for _ in range(3):
        t = time.time(); b0.upload(qs.seqs, qs.offs); t1 = time.time(); b0.sync_upload(); t2 = time.time()
        print("registered" if reg else "pageable", "upload call %.2f ms, arrived after %.2f ms" % (1e3 * (t1 - t), 1e3 * (t2 - t)))
b0.search(); b0.sync(); b1.upload(qs.seqs, qs.offs); b1.sync_upload()
for _ in range(3):
    t = time.time(); b0.search(); b1.upload(qs.seqs, qs.offs); t1 = time.time(); b1.sync_upload(); t2 = time.time(); b0.sync(); t3 = time.time()
    print("beside a search: upload call %.2f ms, arrived after %.2f ms, search done after %.2f ms" % (1e3 * (t1 - t), 1e3 * (t2 - t), 1e3 * (t3 - t)), b0.stats()["ms_total"])
for _ in range(3):
    t = time.time(); b0.search(); b1.search(); b0.sync(); t1 = time.time(); b1.sync(); t2 = time.time()
    print("two searches back to back: first done %.2f ms, second %.2f ms" % (1e3 * (t1 - t), 1e3 * (t2 - t)), b0.stats()["ms_total"], b1.stats()["ms_total"])
for _ in range(3):
    t = time.time(); b0.search(); b0.sync(); t1 = time.time(); h = b0.fetch(reuse=True); t2 = time.time()
    print("search %.2f ms, fetch %.2f ms" % (1e3 * (t1 - t), 1e3 * (t2 - t1)))
