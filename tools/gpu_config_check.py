"""BASELINE.json configurations other than the bench line (C1 small path, one C4 shard: 5M-seq nt DB, C5: 2M-seq
protein DB) at full database size on one GPU: the whole query set runs on the GPU, a sample of it through the
oracle for bit-exact comparison (run on the GPU box; the oracle's single-threaded index build dominates the wall time)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import orc  # noqa: E402
from usearch12_amd import capi, synth  # noqa: E402


def subset(qs, n):
    offs = qs.offs[:n + 1].copy()
    return qs.seqs[:int(offs[-1])], offs


def check(name, db, qs, aa, ident, n_oracle, **kw):
    t0 = time.time()
    p = capi.params(is_nucleo=not aa, id=ident, **kw)
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    t_build = time.time() - t0
    bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
    bat.upload(qs.seqs, qs.offs)
    best = None
    for _ in range(2):
        bat.search(); bat.sync()
        st = bat.stats()
        if best is None or st["ms_total"] < best["ms_total"]:
            best = st
    hits, nh, pool = bat.fetch()
    kh = bat.kernel_hits()
    t1 = time.time()
    odb = orc.OrcDB(orc.params(is_nucleo=not aa, id=ident, **kw), db.seqs, db.offs)
    t_obuild = time.time() - t1
    sq, so = subset(qs, n_oracle)
    oh, onh, opool = odb.search(sq, so, nthreads=min(128, os.cpu_count() or 1))
    t2 = time.time()
    k = int(nh[:n_oracle].sum())
    ok = np.array_equal(nh[:n_oracle], onh) and all(np.array_equal(hits[:k][f], oh[f]) for f in hits.dtype.names if f != "cigar_off")
    if ok:
        for a, b in zip(hits[:k][::53], oh[::53]):
            ok = ok and np.array_equal(pool[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["cigar_len"])],
                                       opool[int(b["cigar_off"]):int(b["cigar_off"]) + int(b["cigar_len"])])
    out = dict(config=name, ok=bool(ok), db_seqs=db.n, queries=qs.n, hits=int(len(hits)), queries_per_s=round(qs.n / (best["ms_total"] * 1e-3)),
               ms_rank=round(best["ms_rank"], 2), ms_align=round(best["ms_align"], 2), ms_rank_setup=round(best["ms_rank_setup"], 2),
               postings_per_query=round(best["postings"] / max(1, qs.n)),
               rank_algorithmic_GBps=round((4.0 * best["postings"] + best["query_letters"]) / (best["ms_rank"] * 1e-3) / 1e9, 1),
               rank_frac_of_hbm_peak=round((4.0 * best["postings"] + best["query_letters"]) / (best["ms_rank"] * 1e-3) / 8e12, 4),
               roofline_k_rank=dict(bound="hbm", algorithmic_bytes=int(4 * best["postings"] + best["query_letters"]), kernel_ms=round(best["ms_rank"], 3),
                                    achieved_GBps=round((4.0 * best["postings"] + best["query_letters"]) / (best["ms_rank"] * 1e-3) / 1e9, 1), peak_GBps=8000.0,
                                    frac=round((4.0 * best["postings"] + best["query_letters"]) / (best["ms_rank"] * 1e-3) / 8e12, 4),
                                    kernels=(kh.get("r2_kernel") or "k_rank") + (" + k_rank over %d deferred units" % kh["deferred"] if kh["r2_launched"] else ""),
                                    ms_bitmap_kernel=round(kh["ms_rank2"], 3), ms_k_rank_behind_it=round(kh["ms_rank_deferred"], 3), units_bitmap_kernel=kh["r2_units"]),
               pairs_aligned=int(best["pairs_aligned"]), index_build_s=round(t_build, 2),
               oracle_checked_queries=n_oracle, oracle_index_build_s=round(t_obuild, 1), oracle_search_s=round(t2 - t1 - t_obuild, 1))
    print(json.dumps(out), flush=True)
    return ok


which = set(sys.argv[1:]) or {"C1", "C4", "C5"}
ok = True
if "C1" in which:
    db = synth.make_db(1, 50_000, 250); qs = synth.make_queries(1, db, 10_000, 250)
    ok &= check("C1 10k x 250nt vs 50k (small path)", db, qs, False, 0.97, 10_000)
if "C4" in which:
    db = synth.make_db(4, 5_000_000, 250); qs = synth.make_queries(4, db, 500_000, 250)
    ok &= check("C4 shard: 500k x 250nt vs 5M", db, qs, False, 0.97, 20_000)
if "C5" in which:
    nq5 = 1_000_000 if "full" in which else 200_000
    db = synth.make_db(5, 2_000_000, 300, aa=True); qs = synth.make_queries(5, db, nq5, 300, aa=True)
    ok &= check("C5 %dk x 300aa vs 2M" % (nq5 // 1000), db, qs, True, 0.8, 10_000)
print("ALL OK" if ok else "FAILED")
