#!/usr/bin/env python3
"""The same query batch through the search N times in one process: CRC of the fetched hit table and run pool per run (determinism of
k_rank / k_align / k_local at full size).  usage: search_repeat.py [--runs 30] [--local] [--id 0.97] [--queries 1000000] [--db 1000000]"""
import argparse
import json
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usearch12_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--runs", type=int, default=30)
ap.add_argument("--queries", type=int, default=1_000_000)
ap.add_argument("--db", type=int, default=1_000_000)
ap.add_argument("--id", type=float, default=0.97)
ap.add_argument("--local", action="store_true")
ap.add_argument("--aa", action="store_true")
ap.add_argument("--both", action="store_true")
a = ap.parse_args()
L = 300 if a.aa else 250
db = synth.make_db(2, a.db, L, aa=a.aa)
qs = synth.make_queries(2, db, a.queries, L, aa=a.aa)
if a.both:
    qs = synth.revcomp_some(2, qs)
kw = dict(local_evalue=1e-6) if a.local else {}
if a.both:
    kw["strand_both"] = 1
gdb = capi.UgsDB(capi.params(is_nucleo=not a.aa, id=a.id, **kw), db.seqs, db.offs, device=0)
bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
bat.upload(qs.seqs, qs.offs)
seen = {}
for k in range(a.runs):
    bat.search(); bat.sync()
    h, nh, pool = bat.fetch()
    crc = zlib.crc32(nh.tobytes())
    for f in h.dtype.names:
        if f != "cigar_off":
            crc = zlib.crc32(h[f].tobytes(), crc)
    # paths in hit order (the pool's layout depends on the order in which waves allocate)
    crc = zlib.crc32(h["cigar_len"].tobytes(), crc)
    seen.setdefault((len(h), crc), []).append(k)
print(json.dumps({"what": ("usearch_local" if a.local else "usearch_global") + " id %g, %d queries vs %d" % (a.id, qs.n, db.n), "runs": a.runs,
                  "distinct_results": len(seen), "results": [[n, c, len(v)] for (n, c), v in seen.items()]}))
sys.exit(0 if len(seen) == 1 else 1)
