"""bitmap ranking kernel (ugs_rank2.hip) against k_rank on the C2 shape: the candidate lists of every unit must be identical;
prints both kernels' times and which code ran.  python tools/r2_check.py [queries] [db_seqs] [aa] [K=V ...]"""
import os, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usearch12_amd import capi, synth

AA = "aa" in sys.argv[1:]            # protein shape (C5): 300 aa, -id 0.8 -> the gather variant k_rank2g
args = [a for a in sys.argv[1:] if "=" not in a and a != "aa"]
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=", 1); os.environ[k] = v
nq = int(args[0]) if len(args) > 0 else 200000
ndb = int(args[1]) if len(args) > 1 else 1000000
seed = 2 if ndb == 1000000 else 4
if AA: seed = 5
db = synth.make_db(seed, ndb, 300 if AA else 250, aa=AA)
qs = synth.make_queries(seed, db, nq, 300 if AA else 250, aa=AA)
res = {}
R3_GIVEN = os.environ.get("UGS_R3")
MODES = ["0", "1"] if (not AA or R3_GIVEN is not None) else ["0", "1:0", "1:1"]       # protein: k_rank, k_rank2g (UGS_R3=0), k_rank3g
for mode in MODES:
    os.environ["UGS_RANK2"] = mode[0]
    if ":" in mode:
        os.environ["UGS_R3"] = mode[2]
    gdb = capi.UgsDB(capi.params(is_nucleo=not AA, id=0.8 if AA else 0.97), db.seqs, db.offs, device=0)
    bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
    bat.upload(qs.seqs, qs.offs)
    r = []
    for _ in range(4):
        bat.search(); bat.sync()
        st = bat.stats(); r.append((st["ms_rank"], st["ms_align"], st["ms_rank_setup"]))
    cand, cnt, n = bat.candidates()
    h, nh, pool = bat.fetch()
    kh = bat.kernel_hits()
    res[mode] = (cand.copy(), cnt.copy(), n.copy(), h.copy())
    print("UGS_RANK2=%s rank %.2f align %.2f setup %.2f ms" % ((mode,) + tuple(np.min(np.array(r[1:]), axis=0))), "hits", len(h),
          "crc", zlib.crc32(h["target"].tobytes()) ^ zlib.crc32(h["ids"].tobytes()), kh, flush=True)
    del bat, gdb
ok = True
for mode in MODES[1:]:
  a, b = res["0"], res[mode]
  print("-- k_rank vs mode", mode)
  if not np.array_equal(a[2], b[2]):
      bad = np.nonzero(a[2] != b[2])[0]
      print("cand_n differs in", len(bad), "units, first", bad[:5], a[2][bad[:5]], b[2][bad[:5]]); ok = False
  K = a[0].shape[1]
  mask = np.arange(K)[None, :] < np.minimum(a[2], b[2])[:, None]
  dc = (a[0] != b[0]) & mask
  dn = (a[1] != b[1]) & mask
  if dc.any() or dn.any():
      bad = np.nonzero((dc | dn).any(axis=1))[0]
      print("candidate lists differ in", len(bad), "units; first:", bad[:5]); ok = False
      for u in bad[:3]:
          m = int(min(a[2][u], b[2][u]))
          print(" unit", u, "n", a[2][u], b[2][u])
          print("  k_rank ", list(zip(a[0][u, :m].tolist(), a[1][u, :m].tolist()))[:12])
          print("  k_rank2", list(zip(b[0][u, :m].tolist(), b[1][u, :m].tolist()))[:12])
  for f in a[3].dtype.names:
      if f not in ("cigar_off",) and not np.array_equal(a[3][f], b[3][f]):          # (path-pool offsets depend on the allocation order)
          print("hit tables differ in field", f); ok = False
print("IDENTICAL" if ok else "DIFFERENT")
