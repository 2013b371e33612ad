#!/usr/bin/env python3
"""Does k_align of one sub-batch overlap k_rank of the next when both kernels leave half of every CU free?
(measure instead of arguing.)  C2 shape, 1M queries per arm:
  full    one stream, one batch of 1M, both kernels at their full residency (the product's step)
  half    the same with both grids halved (UGS_RANK_WGS_PER_CU=2, UGS_ALIGN_WGS_PER_CU=2: 2 + 2 workgroups fit a CU) - what
          each kernel loses by itself at half residency
  corun   two handles on the device (= two streams, index replicated), halved grids, the 1M queries as sub-batches
          alternating between the streams with stream B started half a sub-batch late, so that one stream's k_align runs
          beside the other's k_rank; everything enqueued before the first sync
  corun_full  the same two streams at full residency (the hardware decides what co-resides)
Prints one JSON line per arm (wall ms for the 1M queries, kernel ms summed from the batches' own HIP events).  Run under
`rocprofv3 --kernel-trace` for the timeline (tools/overlap_timeline.py summarises the trace)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usearch12_amd import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--db", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=1_000_000)
    ap.add_argument("--subs", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    db = synth.make_db(2, args.db, 250)
    parts = [synth.make_queries(2 + 104729 * c, db, 250_000, 250) for c in range((args.queries + 249_999) // 250_000)]
    seqs = np.concatenate([q.seqs for q in parts])
    lens = np.concatenate([np.diff(q.offs.astype(np.int64)) for q in parts])[:args.queries]
    offs = np.zeros(len(lens) + 1, np.uint64); offs[1:] = np.cumsum(lens).astype(np.uint64)
    seqs = np.ascontiguousarray(seqs[:int(offs[-1])])
    capi._chk(capi.lib().ugs_host_register(seqs.ctypes.data, seqs.nbytes))
    p = capi.params(is_nucleo=True, id=0.97)
    dbs = [capi.UgsDB(p, db.seqs, db.offs, device=0) for _ in range(2)]

    def sub(lo, hi):
        return seqs[int(offs[lo]):int(offs[hi])], offs[lo:hi + 1] - offs[lo]

    def setenv(half):
        for k in ("UGS_RANK_WGS_PER_CU", "UGS_ALIGN_WGS_PER_CU"):
            os.environ.pop(k, None)
            if half:
                os.environ[k] = "2"

    def arm(name, half, plan):
        """plan: list of (stream index, lo, hi) in enqueue order"""
        setenv(half)
        bats = []
        for s, lo, hi in plan:
            b = capi.UgsBatch(dbs[s], hi - lo, int(offs[hi] - offs[lo]) + 64)
            b.upload(*sub(lo, hi))
            bats.append(b)
        best = None
        for rep in range(args.reps + 1):
            t0 = time.time()
            for b in bats:
                b.search()
            for b in bats:
                b.sync()
            ms = (time.time() - t0) * 1e3
            st = [b.stats() for b in bats]
            rec = {"arm": name, "wall_ms": ms, "ms_rank_sum": sum(s["ms_rank"] for s in st), "ms_align_sum": sum(s["ms_align"] for s in st),
                   "ms_setup_sum": sum(s["ms_rank_setup"] for s in st), "hits": int(sum(s["hits"] for s in st)), "sub_batches": len(plan),
                   "half_grids": bool(half)}
            if rep and (best is None or ms < best["wall_ms"]):
                best = rec
        print(json.dumps(best), flush=True)
        for b in bats:
            b.close() if hasattr(b, "close") else None
        return best

    n = args.queries
    arm("full", False, [(0, 0, n)])
    arm("half", True, [(0, 0, n)])
    k = args.subs
    step = n // k
    # stream A: sub-batches 0, 2, 4 ...; stream B: a half sub-batch first, then 1, 3, ... (minus that half at the end)
    cuts = [0]
    plan = []
    lo = 0
    hb = step // 2
    plan.append((1, 0, hb)); lo = hb                       # B's late start
    s = 0
    while lo < n:
        hi = min(n, lo + step)
        plan.append((s, lo, hi))
        s ^= 1
        lo = hi
    arm("corun", True, plan)
    arm("corun_full", False, plan)
    arm("serial_subs", False, [(0, a, b) for (_, a, b) in plan])


if __name__ == "__main__":
    main()
