#!/usr/bin/env python3
"""Throughput of the gapped x-drop kernel on one GPU: N anchored extensions of
L-letter reads against mutated copies.  Prints one JSON line: jobs/s and DP cells/s from the kernel's
own HIP events, the end-to-end rate of the host-buffer ABI call, and the oracle's single-thread rate on
a sample of the same jobs (the CPU baseline; test infrastructure, not part of the product)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from usearch12_amd import capi  # noqa: E402
from usearch12_amd.abi import XDROP_ALIGN, XDROP_JOB_DTYPE  # noqa: E402


def make(seed, n, L, aa, p_sub=0.03, p_indel=0.005):
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer((b"ACDEFGHIKLMNPQRSTVWY" if aa else b"ACGT"), np.uint8)
    a = alpha[rng.integers(0, len(alpha), (n, L))]
    bs, jobs = [], np.zeros(n, XDROP_JOB_DTYPE)
    k = 4 if aa else 12
    for r in range(n):
        row = a[r]
        m = rng.random(L)
        keep = m >= p_indel
        sub = (m >= p_indel) & (m < p_indel + p_sub)
        b = row.copy()
        b[sub] = alpha[rng.integers(0, len(alpha), int(sub.sum()))]
        pos = int(rng.integers(L // 4, L // 2))
        keep[pos:pos + k] = True
        b[pos:pos + k] = row[pos:pos + k]
        bb = b[keep]
        ins = rng.random(len(bb)) < p_indel
        ins[max(0, int(keep[:pos].sum()) - 1):int(keep[:pos].sum()) + k + 1] = False
        bb = np.insert(bb, np.nonzero(ins)[0], alpha[rng.integers(0, len(alpha), int(ins.sum()))])
        # anchor = the protected exact k-mer
        jpos = int(keep[:pos].sum()) + int(ins[:int(keep[:pos].sum())].sum())
        assert bytes(bb[jpos:jpos + k]) == bytes(row[pos:pos + k])
        bs.append(bb)
        jobs[r] = (r, r, pos, jpos, k, XDROP_ALIGN)
    a_offs = np.arange(n + 1, dtype=np.uint64) * L
    b_offs = np.zeros(n + 1, np.uint64)
    b_offs[1:] = np.cumsum([len(x) for x in bs])
    return (a.reshape(-1).copy(), a_offs), (np.concatenate(bs), b_offs), jobs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=200000)
    ap.add_argument("--len", type=int, default=250)
    ap.add_argument("--aa", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=3000)
    args = ap.parse_args()
    A, B, jobs = make(5, args.jobs, args.len, args.aa)
    p = capi.xdrop_params(not args.aa)
    best_ms, best_wall, cells = 1e30, 1e30, 0
    for _ in range(args.reps):
        t0 = time.time()
        hsps, pool = capi.xdrop_batch(p, A, B, jobs)
        wall = time.time() - t0
        ms, cells = capi.xdrop_last_stats()
        best_ms, best_wall = min(best_ms, ms), min(best_wall, wall)
    out = dict(metric="xdrop_jobs_per_s", jobs=args.jobs, read_len=args.len, aa=args.aa, ms_kernel=round(best_ms, 3),
               jobs_per_s_kernel=round(args.jobs / (best_ms * 1e-3)), gcells_per_s=round(cells / (best_ms * 1e-3) / 1e9, 2),
               cells_per_job=round(cells / args.jobs, 1), jobs_per_s_abi_host_buffers=round(args.jobs / best_wall),
               mean_score=float(hsps["score"].mean()), mean_runs=float(hsps["path_len"].mean()))
    try:
        import orc
        po = orc.xdrop_params(not args.aa, 32.0)
        n = min(args.cpu_sample, args.jobs)
        a_s, a_o = A
        b_s, b_o = B
        seqs = [(bytes(a_s[int(a_o[k]):int(a_o[k + 1])]), bytes(b_s[int(b_o[k]):int(b_o[k + 1])])) for k in range(n)]
        t0 = time.time()
        ok = 0
        for k in range(n):
            r = orc.xdrop_job(po, seqs[k][0], seqs[k][1], XDROP_ALIGN, (int(jobs[k]["anc_loi"]), int(jobs[k]["anc_loj"]), int(jobs[k]["anc_len"])))
            ok += (r[0] == float(hsps[k]["score"]))
        dt = time.time() - t0
        out["cpu_oracle_jobs_per_s_1thread"] = round(n / dt)
        out["cpu_sample_agree"] = "%d/%d" % (ok, n)
    except Exception as e:  # oracle not built
        out["cpu_oracle"] = "unavailable: %s" % e
    print(json.dumps(out))


if __name__ == "__main__":
    main()
