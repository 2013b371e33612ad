"""k_align split: 100 k C2-shape queries that all have a relative in the DB / that are all random (phase clocks with UGS_LIB = a -DUGS_ALIGN_CLOCKS=1|2|3 build, tools/build_variant_of.sh ugs_align)"""
import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
os.environ["UGS_PHASE_CLOCKS"] = "1"
from usearch12_amd import capi, synth
AA = os.environ.get("RQ_SHAPE", "") == "aa"          # RQ_SHAPE=aa: the C5 shape
if AA:
    db = synth.make_db(5, 2000000, 300, aa=True)
else:
    db = synth.make_db(2, 1000000, 250)
gdb = capi.UgsDB(capi.params(is_nucleo=not AA, id=0.8 if AA else 0.97), db.seqs, db.offs, device=0)
for fr in (0.0, 1.0):
    qs = synth.make_queries(5 if AA else 2, db, 100000, 300 if AA else 250, aa=AA, frac_random=fr)
    bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
    bat.upload(qs.seqs, qs.offs)
    for _ in range(2):
        bat.search(); bat.sync()
    print("frac_random", fr, flush=True)
    st = bat.stats()
    print("  align %.2f ms pairs %d hits %d dp_cells %d" % (st["ms_align"], st["pairs_aligned"], st["hits"], st["dp_cells"]), flush=True)
