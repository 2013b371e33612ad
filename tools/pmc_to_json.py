#!/usr/bin/env python3
"""gpurun_out/<tag>_pmc_pass*.txt (tools/prof_round.sh) -> profiles/<tag>_pmc.json in the schema bench.py reads:
traffic per launch = FETCH_SIZE KiB x 2 (MI355X_MICROARCH.md: gfx950 reports half of a coalesced read stream) + WRITE_SIZE KiB.
usage: pmc_to_json.py <tag> <db_seqs> <queries> "<note>" """
import glob
import json
import re
import sys

tag, db_seqs, queries, note = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
# the build the passes ran on: the source hash the bench line of the same gpurun call carries (usearch12_amd/build.py csrc_hash);
# bench.py attaches this file's traffic to its roofline only when the running build has the same hash
try:
    csrc = json.loads(open("gpurun_out/%s_bench_c2.json" % tag).read().strip().splitlines()[-1])["detail"]["csrc_sha16"]
except Exception:
    csrc = None
raw, mix = {}, {}
for f in sorted(glob.glob("gpurun_out/%s_pmc_pass*.txt" % tag)):
    for ln in open(f):
        m = re.match(r"^(k_\w+)\s+(\w+)\s+([\d.e+]+)\s+\(mean of (\d+) launches\)", ln)
        if not m:
            continue
        k, c, v = m.group(1), m.group(2), float(m.group(3))
        if c in ("FETCH_SIZE", "WRITE_SIZE"):
            raw.setdefault(k, {})[c + "_KiB"] = v
        else:
            mix.setdefault(k, {})[c] = v
traffic = {k: int((2 * d.get("FETCH_SIZE_KiB", 0) + d.get("WRITE_SIZE_KiB", 0)) * 1024) for k, d in raw.items()}
json.dump({"db_seqs": db_seqs, "queries": queries, "csrc_sha16": csrc, "note": note, "raw": raw, "traffic_bytes_per_launch": traffic,
           "instruction_mix_per_launch": mix, "valu_issue_peak_inst_per_s": 256 * 4 * 2.4e9 / 2,
           "valu_issue_note": "MI355X_MICROARCH.md: 4 SIMD-32 per CU, 2 cycles per wave64 VALU instruction at 2.4 GHz; measured per instruction in "
                              "profiles/r03c_ubench_issue_cost.jsonl: only add/sub/and/or/xor/lshrrev/mov reach ~2.4 cycles, the other integer ops ~4.3"},
          open("profiles/%s_pmc.json" % tag, "w"), indent=1)
print(json.dumps(traffic))
