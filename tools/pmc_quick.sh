#!/bin/bash
# usage: tools/pmc_quick.sh <tag> <nq> <counter> [...]  - one rocprofv3 --pmc pass over tools/rank_quick.py (GPU box); env UGS_LIB etc. pass through
set -e
tag=$1; nq=$2; shift; shift
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out/pmcq_$tag"
cd /tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$REPO/gpurun_out/pmcq_$tag" -- python "$REPO/tools/rank_quick.py" $nq > "$REPO/gpurun_out/pmcq_$tag/run.log" 2>&1 || { tail -5 "$REPO/gpurun_out/pmcq_$tag/run.log"; exit 1; }
cd "$REPO"
python - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
files = glob.glob("gpurun_out/pmcq_%s/**/*counter_collection.csv" % tag, recursive=True)
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if not k.startswith(("k_rank", "k_align")):
            continue
        acc[(k, row["Counter_Name"])] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
for (k, c), v in sorted(acc.items()):
    print("%s %-14s %-24s %.6g (per launch, %d launches)" % (tag, k, c, v / n[(k, c)], n[(k, c)]))
PY
