"""k_rank3g on the C5 shape: ranking time against the postings aimed at per super-partition (UGS_R3_PPS) and k_rank2g beside it.
python tools/r3_sweep.py [queries] [db_seqs] [pps ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usearch12_amd import capi, synth

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
ndb = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
pps = [a for a in sys.argv[3:] if "=" not in a] or ["1024", "1536", "2048", "3072", "4096"]
for a in sys.argv[3:]:
    if "=" in a:
        k, v = a.split("=", 1); os.environ[k] = v
db = synth.make_db(5, ndb, 300, aa=True)
qs = synth.make_queries(5, db, nq, 300, aa=True)
ref = None
for mode in ["r2g"] + pps:
    os.environ.pop("UGS_R3_PPS", None)
    os.environ["UGS_R3"] = "0" if mode == "r2g" else "1"
    if mode != "r2g":
        os.environ["UGS_R3_PPS"] = mode
    gdb = capi.UgsDB(capi.params(is_nucleo=False, id=0.8), db.seqs, db.offs, device=0)
    bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
    bat.upload(qs.seqs, qs.offs)
    r = []
    for _ in range(5):
        bat.search(); bat.sync()
        st = bat.stats(); r.append(st["ms_rank"])
    kh = bat.kernel_hits()
    cand, cnt, n = bat.candidates()
    sig = (cand.tobytes(), cnt.tobytes(), n.tobytes())
    if ref is None:
        ref = sig
    print("%-6s ranking %.3f ms (min of 4)  %s deferred %d  %s" % (mode, min(r[1:]), kh["r2_kernel"], kh["deferred"], "same" if sig == ref else "DIFFERENT"), flush=True)
    del bat, gdb
