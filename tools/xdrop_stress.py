#!/usr/bin/env python3
"""Determinism stress of the gapped x-drop kernels: the same seeded batch through ugs_xdrop_batch (k_xdrop) N times
in one process, every run compared with the first and with the oracle (checker only); then the same for one
usearch_local search (k_rank + k_local).  Prints one JSON line per batch:
  {"what":..., "iters":N, "runs_differing_from_first":.., "runs_differing_from_oracle":.., "first_diffs":[...]}
UGS_LIB selects a variant build (tools/build_xd_variant.sh)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402  (checker)
import test_gpu_xdrop as tx  # noqa: E402
from usearch12_amd import capi, synth  # noqa: E402
from usearch12_amd.abi import XDROP_FWD, XDROP_JOB_DTYPE, path_text  # noqa: E402


def xdrop(seed, aa, x, n, lmin, lmax, iters, no_oracle=False):
    cases = tx._random_batch(seed, aa, n, lmin, lmax)
    want = None
    if not no_oracle:
        p = orc.xdrop_params(not aa, x)
        want = []
        for (mode, a, b, anc) in cases:
            o = orc.xdrop_job(p, a, b, mode, anc)
            want.append(o[:6] if mode != XDROP_FWD else (o[0], 0, 0, o[3], o[4], o[5]))
    jobs = np.zeros(len(cases), XDROP_JOB_DTYPE)
    for k, (mode, a, b, anc) in enumerate(cases):
        jobs[k] = (k, k, anc[0], anc[1], anc[2], mode)
    gp = capi.xdrop_params(not aa, xdrop=x)
    A, B = tx.pack([c[1] for c in cases]), tx.pack([c[2] for c in cases])

    def decode(hsps, pool):
        return [(float(h["score"]), int(h["loi"]), int(h["loj"]), int(h["leni"]), int(h["lenj"]),
                 path_text(pool, h["path_off"], h["path_len"])) for h in hsps]
    first = first_raw = None
    d_first = d_orc = 0
    diffs = []
    t0 = time.time()
    for it in range(iters):
        hsps, pool = capi.xdrop_batch(gp, A, B, jobs)
        raw = (hsps.tobytes(), pool.tobytes())
        if first_raw is None:
            first_raw = raw
            first = decode(hsps, pool)
            d_orc += bool(want is not None and first != want)
            if want is not None and first != want:
                diffs.append({"iter": 0, "jobs_differing_from_oracle": [k for k in range(len(cases)) if first[k] != want[k]][:10]})
            continue
        if raw == first_raw:
            continue
        got = decode(hsps, pool)
        bad_f = [k for k in range(len(cases)) if got[k] != first[k]]
        bad_o = [k for k in range(len(cases)) if want is not None and got[k] != want[k]]
        d_first += 1
        d_orc += bool(bad_o)
        for k in (bad_o or bad_f)[:3]:
            if len(diffs) < 12:
                ref = want[k] if want is not None else first[k]
                h = hsps[k]
                runs = pool[int(h["path_off"]):int(h["path_off"]) + int(h["path_len"])]
                diffs.append({"iter": it, "job": k, "mode": int(cases[k][0]), "la": len(cases[k][1]), "lb": len(cases[k][2]),
                              "anc": list(map(int, cases[k][3])), "got5": list(got[k][:5]), "want5": list(ref[:5]),
                              "got_runs": ["%d%s" % (r >> 2, "MDI?"[r & 3]) for r in runs.tolist()][:80],
                              "want_runs": compress(ref[5])[:80]})
    print(json.dumps({"what": "k_xdrop seed=%d aa=%d x=%g n=%d" % (seed, aa, x, n), "lib": os.path.basename(capi.LIB_PATH), "iters": iters,
                      "runs_differing_from_first": d_first, "runs_differing_from_oracle": d_orc, "s": round(time.time() - t0, 1),
                      "first_diffs": diffs}), flush=True)
    return d_first + d_orc


def compress(path):
    out, k = [], 0
    while k < len(path):
        j = k
        while j < len(path) and path[j] == path[k]:
            j += 1
        out.append("%d%s" % (j - k, path[k]))
        k = j
    return out


def local(iters, nq=4000, ndb=20000):
    db = synth.make_db(11, ndb, 400)
    qs = synth.make_queries(11, db, nq, 400)
    kw = dict(id=None, local_evalue=1e-6)
    g = capi.UgsDB(capi.params(is_nucleo=True, **kw), db.seqs, db.offs, device=0)
    oh, onh, opool = orc.OrcDB(orc.params(is_nucleo=True, **kw), db.seqs, db.offs).search(qs.seqs, qs.offs)

    def key(h, nh, pool):
        out = [nh.tobytes()]
        for f in h.dtype.names:
            if f != "cigar_off":
                out.append(h[f].tobytes())
        for r in h:
            out.append(pool[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["cigar_len"])].tobytes())
        return b"".join(out)
    want = key(oh, onh, opool)
    first = None
    d_first = d_orc = 0
    t0 = time.time()
    for it in range(iters):
        k = key(*g.search(qs.seqs, qs.offs))
        if first is None:
            first = k
        d_first += k != first
        d_orc += k != want
    print(json.dumps({"what": "usearch_local nq=%d ndb=%d hsps=%d" % (nq, ndb, len(oh)), "lib": os.path.basename(capi.LIB_PATH), "iters": iters,
                      "runs_differing_from_first": d_first, "runs_differing_from_oracle": d_orc, "s": round(time.time() - t0, 1)}), flush=True)
    return d_first + d_orc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--local-iters", type=int, default=20)
    ap.add_argument("--quick", action="store_true", help="seed-5 batch only")
    ap.add_argument("--dirty", type=float, default=0.0, help="GiB of HBM filled with random bytes and freed first, so that the library's "
                    "uninitialised scratch starts from garbage as it does late in a long test session")
    args = ap.parse_args()
    if args.dirty > 0:
        import torch
        n = int(args.dirty * (1 << 30)) // 8
        chunks = [torch.randint(-(1 << 62), 1 << 62, (min(n, 1 << 27),), dtype=torch.int64, device="cuda") for _ in range(max(1, n >> 27))]
        torch.cuda.synchronize()
        del chunks
        torch.cuda.empty_cache()
    bad = xdrop(5, False, 200.0, 1500, 5, 900, args.iters)
    if not args.quick:
        bad += xdrop(1, False, 32.0, 1500, 5, 900, max(5, args.iters // 5))
        bad += xdrop(4, True, 60.0, 1500, 5, 900, max(5, args.iters // 5))
    if args.local_iters:
        bad += local(args.local_iters)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
