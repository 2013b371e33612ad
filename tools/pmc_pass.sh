#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counter> [<counter> ...]   - one rocprofv3 --pmc pass over one bench step (GPU box)
# writes gpurun_out/pmc_<tag>/ and prints per-kernel sums of each counter
set -e
tag=$1; shift
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out/pmc_$tag"
cd /tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$REPO/gpurun_out/pmc_$tag" -- python "$REPO/bench.py" --steps 1 --warmup 0 --cpu-baseline none --other-configs none --emulate-world 0 > "$REPO/gpurun_out/pmc_$tag/bench.log" 2>&1 || { tail -5 "$REPO/gpurun_out/pmc_$tag/bench.log"; exit 1; }
cd "$REPO"
python - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
files = glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % tag, recursive=True)
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if not k.startswith(("k_rank", "k_align")):
            continue
        acc[(k, row["Counter_Name"])] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
for (k, c), v in sorted(acc.items()):          # per launch (bench.py launches each kernel once per step plus once while priming)
    print("%-14s %-28s %.6g  (mean of %d launches)" % (k, c, v / n[(k, c)], n[(k, c)]))
PY
