#!/bin/bash
# usage: tools/pmc_c5.sh <tag> <counter> [...]  - one rocprofv3 --pmc pass over the protein shape (tools/r3_sweep.py: k_rank2g once, k_rank3g
# once, 200 k queries vs 2 M sequences); prints per-kernel counter sums per launch (GPU box)
set -e
tag=$1; shift
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out/pmc5_$tag"
cd /tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$REPO/gpurun_out/pmc5_$tag" -- python "$REPO/tools/r3_sweep.py" 200000 2000000 4096 > "$REPO/gpurun_out/pmc5_$tag/run.log" 2>&1 || { tail -5 "$REPO/gpurun_out/pmc5_$tag/run.log"; exit 1; }
cd "$REPO"
python - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
files = glob.glob("gpurun_out/pmc5_%s/**/*counter_collection.csv" % tag, recursive=True)
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if not k.startswith(("k_rank2g", "k_rank3g", "k_align")):
            continue
        acc[(k, row["Counter_Name"])] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
for (k, c), v in sorted(acc.items()):
    print("%s %-14s %-24s %.6g (per launch, %d launches)" % (tag, k, c, v / n[(k, c)], n[(k, c)]))
PY
rm -rf "$REPO/gpurun_out/pmc5_$tag"
