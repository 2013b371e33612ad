"""Where a strong-scaling step's time goes ON THE GPU: from a rocprofv3 --kernel-trace --memory-copy-trace run of
`bench.py --cpu-baseline none --other-configs none --parity-sample 0` (csv output), the activities of a few consecutive steps of the proxy
phase (the steps whose k_rank2 launch is short: 1/N of the batch) with their start offsets, durations and the idle gaps between them.
  python tools/proxy_timeline.py <dir with *_kernel_trace.csv / *_memory_copy_trace.csv> [out.txt]"""
import csv
import glob
import os
import sys


def load(d):
    acts = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acts.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"][:70], r.get("Queue_Id", "?")))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acts.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "copy")[:40], "-"))
    acts.sort()
    return acts


def main():
    acts = load(sys.argv[1])
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    r2 = [(s, e) for s, e, k, n, q in acts if k == "K" and "k_rank2" in n]
    if not r2:
        print("no k_rank2 launches in the trace", file=out)
        return
    full = max(e - s for s, e in r2)
    small = [(s, e) for s, e in r2 if (e - s) < 0.25 * full]
    print("k_rank2 launches: %d, longest %.2f ms, proxy-sized (< 1/4 of it): %d" % (len(r2), full / 1e6, len(small)), file=out)
    if len(small) < 8:
        return
    # three consecutive proxy steps from the middle of the phase
    i0 = len(small) // 2
    t0, t1 = small[i0][0], small[i0 + 3][0]
    print("window: three steps, %.3f ms per step" % ((t1 - t0) / 3e6), file=out)
    last_end = None
    busy = 0
    for s, e, k, n, q in acts:
        if e < t0 or s >= t1:
            continue
        gap = "" if last_end is None or k != "K" else "  gap %+7.1f us" % ((s - last_end) / 1e3)
        print("%9.3f ms  %s %8.1f us  q%s  %s%s" % ((s - t0) / 1e6, k, (e - s) / 1e3, q, n, gap), file=out)
        if k == "K":
            busy += min(e, t1) - max(s, t0)
            last_end = e if last_end is None else max(last_end, e)
    print("kernel-busy time in the window: %.3f ms of %.3f ms" % (busy / 1e6, (t1 - t0) / 1e6), file=out)


if __name__ == "__main__":
    main()
