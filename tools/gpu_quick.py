"""Quick GPU bring-up: run golden cases through the HIP path, print first mismatches."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import golden_util as G
import orc
from usearch12_amd import capi

names = sys.argv[1:] or G.case_names()
bad = 0
for name in names:
    c, db, qs, b6, uc = G.load(name)
    p = capi.params(is_nucleo=not c["aa"], id=c["id"], **G.params_kw(c))
    t = time.time()
    try:
        gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
        t1 = time.time()
        bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
        bat.upload(qs.seqs, qs.offs); bat.search(); bat.sync()
        hits, nh, pool = bat.fetch()
        st = bat.stats()
    except Exception as e:
        print(name, "EXC", e); bad += 1; continue
    t2 = time.time()
    qlens = np.diff(qs.offs.astype(np.int64))
    gb6, guc = orc.format_outputs(capi.lib(), "ugs", hits, nh, pool, qs.labels(), qlens, db.labels(), not c["aa"])
    ok = gb6 == b6 and guc == uc
    print(name, "OK" if ok else "MISMATCH", "hits", len(hits), "/", c["n_hits"], "db %.2fs search %.3fs" % (t1 - t, t2 - t1),
          "rank %.2fms align %.2fms" % (st["ms_rank"], st["ms_align"]))
    if not ok:
        bad += 1
        a, b = uc.splitlines(), guc.splitlines()
        k = 0
        for x, y in zip(a, b):
            if x != y:
                print("  REF", x); print("  GPU", y); k += 1
                if k >= 4: break
        # candidate comparison for the first bad query
        op = orc.params(is_nucleo=not c["aa"], id=c["id"], **G.params_kw(c))
        odb = orc.OrcDB(op, db.seqs, db.offs)
        cand, cnt, n = bat.candidates()
        ns = 2 if p.strand_both else 1
        shown = 0
        for qi in range(qs.n):
            q = qs.seqs[int(qs.offs[qi]):int(qs.offs[qi + 1])]
            for s in range(ns):
                qq = q if s == 0 else orc.revcomp(q)
                on, oc, occ = odb.rank(qq, cap=cand.shape[1])
                u = qi * ns + s
                m = min(on, cand.shape[1])
                if n[u] != m or not np.array_equal(cand[u, :m], oc[:m]) or not np.array_equal(cnt[u, :m], occ[:m]):
                    print("  cand mismatch q", qi, "strand", s, "gpu n", n[u], "orc n", on)
                    print("   gpu", list(zip(cand[u, :min(n[u], 8)], cnt[u, :min(n[u], 8)])))
                    print("   orc", list(zip(oc[:8], occ[:8])))
                    shown += 1
                    break
            if shown >= 2: break
        if shown == 0: print("  candidates all equal -> alignment-stage mismatch")
print("BAD", bad)
