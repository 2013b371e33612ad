#!/usr/bin/env python3
"""Is host heap memory written after ugs_xdrop_batch returns?  (Round-3 root-cause hunt for the 'wrong path' of
test_xdrop_matches_oracle[False-200.0-5]: the differing character was 'M' -> 'L' at an 8-byte aligned offset of a
malloc'ed Python string - a 64-bit decrement of freed-and-reused host memory, not anything a kernel computed.)

Every iteration: one ugs_xdrop_batch call on the seed-5 batch, then ~1000 fresh malloc'ed canaries of assorted sizes
(bytearrays filled with 0x4D), a short wait, and a check that every byte is still 0x4D - also of the previous
iteration's canaries.  Prints one JSON line with every corruption found: (iteration, canary size, offset, bytes).
UGS_LIB selects the build (tools/build_xd_variant.sh: UGS_XD_HOSTMODE = which of stream / events are per call)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_xdrop as tx  # noqa: E402
from usearch12_amd import capi  # noqa: E402
from usearch12_amd.abi import XDROP_JOB_DTYPE  # noqa: E402

import ctypes  # noqa: E402
LIBC = ctypes.CDLL(None)
LIBC.malloc.restype = ctypes.c_void_p
LIBC.malloc.argtypes = [ctypes.c_size_t]
LIBC.free.argtypes = [ctypes.c_void_p]
LIBC.memset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
# every malloc bin up to 2 KiB ten times over (a thread's tcache holds 7 chunks per bin), then coarser steps
SIZES = [n for n in range(24, 2048, 16) for _ in range(10)] + [n for n in range(2048, 16384, 128) for _ in range(3)]


def make():
    out = []
    for n in SIZES:
        p = LIBC.malloc(n)
        LIBC.memset(p, 0x4D, n)
        out.append((p, n))
    return out


def release(cans):
    for p, n in cans:
        LIBC.free(p)


def check(cans, it, found):
    for p, n in cans:
        c = ctypes.string_at(p, n)
        if c.count(b"M") != n:
            bad = [k for k in range(n) if c[k] != 0x4D]
            found.append({"iter": it, "size": n, "offsets": bad[:8], "bytes": [c[k] for k in bad[:8]]})
            LIBC.memset(p, 0x4D, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=4000)
    ap.add_argument("--wait-ms", type=float, default=1.0)
    ap.add_argument("--mode", default="xdrop", choices=["xdrop", "search"])
    args = ap.parse_args()
    cases = tx._random_batch(5, False, 1500, 5, 900)
    jobs = np.zeros(len(cases), XDROP_JOB_DTYPE)
    for k, (mode, a, b, anc) in enumerate(cases):
        jobs[k] = (k, k, anc[0], anc[1], anc[2], mode)
    gp = capi.xdrop_params(True, xdrop=200.0)
    A, B = tx.pack([c[1] for c in cases]), tx.pack([c[2] for c in cases])
    if args.mode == "search":
        from usearch12_amd import synth
        db = synth.make_db(11, 20000, 400)
        qs = synth.make_queries(11, db, 2000, 400)
        g = capi.UgsDB(capi.params(is_nucleo=True, id=0.97), db.seqs, db.offs, device=0)
    found, prev = [], []
    first = None
    t0 = time.time()
    for it in range(args.iters):
        if args.mode == "xdrop":
            hsps, pool = capi.xdrop_batch(gp, A, B, jobs)
            raw = hsps.tobytes() + pool.tobytes()
        else:
            hits, nh, pool = g.search(qs.seqs, qs.offs)      # ugs_search_batch: a batch object (events, copy stream) per call
            raw = nh.tobytes()
        cans = make()
        if args.wait_ms:
            time.sleep(args.wait_ms * 1e-3)
        check(cans, it, found)
        check(prev, it, found)
        release(prev)
        prev = cans
        if first is None:
            first = raw
        elif raw != first:
            found.append({"iter": it, "result_differs": True})
    print(json.dumps({"what": "heap canaries after " + args.mode, "lib": os.path.basename(capi.LIB_PATH), "iters": args.iters,
                      "corruptions": len(found), "s": round(time.time() - t0, 1), "found": found[:20]}), flush=True)
    sys.exit(1 if found else 0)


if __name__ == "__main__":
    main()
