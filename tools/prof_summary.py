"""Condense a rocprofv3 *_kernel_stats.csv into a short table (long template names trimmed)."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-60s %6s %14s %14s %8s" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    name = r["Name"]
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    m = re.search(r"(radix_sort_onesweep_iteration|radix_sort_onesweep_global_offsets|partition_impl|transform_impl|init_lookback_scan_state_kernel)", name)
    if m: name = "rocprim::" + m.group(1)
    print("%-60s %6s %14.3f %14.3f %8s" % (name[:60], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"]))
