#!/usr/bin/env python3
"""cluster_fast on the same reads N times in one process: cluster count and a CRC of every result array per run (determinism of the
device loop at full size).  usage: cluster_repeat.py [--reads 5000000] [--species 50000] [--runs 8]"""
import argparse
import json
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usearch12_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=5_000_000)
ap.add_argument("--species", type=int, default=50_000)
ap.add_argument("--runs", type=int, default=8)
a = ap.parse_args()
r = synth.make_reads(3, a.reads, n_species=a.species)
p = capi.cluster_params(0.97)
seen = {}
for k in range(a.runs):
    res = capi.UgsCluster(p, r.seqs, r.offs)
    crc = 0
    for f in ("uniq_cluster", "uniq_nhits", "centroid_uniq", "cluster_size"):
        crc = zlib.crc32(getattr(res, f).tobytes(), crc)
    for f in ("target", "ids", "aln_len", "qlo", "qhi", "tlo", "thi"):
        crc = zlib.crc32(res.hits[f].tobytes(), crc)
    st = res.stats
    print(json.dumps({"run": k, "n_clusters": int(res.n_clusters), "crc": crc, "batches": st.batches, "redone": st.queries_redone,
                      "pairs_frozen": st.pairs_frozen, "inbatch_entries": st.inbatch_entries, "hits": int(len(res.hits))}), flush=True)
    seen.setdefault((int(res.n_clusters), crc), []).append(k)
    res.close()
print(json.dumps({"distinct_results": len(seen), "runs": a.runs}))
sys.exit(0 if len(seen) == 1 else 1)
