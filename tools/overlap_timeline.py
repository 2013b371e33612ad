#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of tools/overlap_probe.py: for every burst of kernels (bursts are separated by
> 20 ms of idle GPU = the arms and their repetitions) the wall span, the time some k_rank runs, the time some k_align
runs and the time BOTH run at once.  usage: overlap_timeline.py <dir with *_kernel_trace.csv> > summary.json"""
import csv
import glob
import json
import sys


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def length(iv):
    return sum(b - a for a, b in iv)


def inter(x, y):
    i = j = 0
    tot = 0
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if b > a:
            tot += b - a
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return tot


def main():
    rows = []
    for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            kind = "k_rank" if "k_rank" in name and "setup" not in name else ("k_align" if "k_align" in name else None)
            if kind:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
    rows.sort()
    bursts, cur = [], []
    for r in rows:
        if cur and r[0] - max(x[1] for x in cur) > 20_000_000:
            bursts.append(cur); cur = []
        cur.append(r)
    if cur:
        bursts.append(cur)
    out = []
    for b in bursts:
        rk = union([(a, e) for a, e, k in b if k == "k_rank"])
        al = union([(a, e) for a, e, k in b if k == "k_align"])
        out.append({"kernels": len(b), "span_ms": (max(x[1] for x in b) - b[0][0]) / 1e6, "k_rank_busy_ms": length(rk) / 1e6,
                    "k_align_busy_ms": length(al) / 1e6, "both_ms": inter(rk, al) / 1e6})
    print(json.dumps({"bursts": out}, indent=1))


if __name__ == "__main__":
    main()
