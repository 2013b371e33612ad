#!/usr/bin/env python3
"""Fixtures for the writers beyond blast6/uc, produced by the UNMODIFIED reference
(oracle/_ref/usearch12) on two of the seeded golden cases: -userout with every supported -userfields name,
-output_no_hits, -matched/-notmatched, -dbmatched/-dbnotmatched, and the -maxhits / -top_hits_only /
-top_hit_only hit-count rules.  Files: tests/golden/out_<run>.<kind>; outputs above 100 kB are kept as their first 20
lines (out_<run>.<kind>.head) plus sha256 and line count in out_manifest.json.  Runs only where /root/reference exists."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

FIELDS = ("query+target+clusternr+evalue+id+fractid+dist+mid+pctpv+pctgaps+pairs+gaps+allgaps+qlo+qhi+tlo+thi+qlor+qhir+tlor+thir+"
          "qlot+qhit+qunt+tlot+thit+tunt+pv+ql+tl+qs+ts+alnlen+opens+exts+raw+bits+aln+caln+qseq+tseq+qstrand+tstrand+qrow+trow+"
          "qrowdots+trowdots+qframe+tframe+orflo+orfhi+orfframe+mism+ids+qcov+tcov+diffs+diffsa+editdiffs")
NOHIT_FIELDS = "query+target+id+ql+clusternr+qseq+alnlen+caln+qstrand"
# run name -> (golden case it reuses, extra reference options, kinds of output files to keep)
RUNS = {
    "nt_all":   ("hard_acc_s", ["-userfields", FIELDS], ["user", "b6", "uc", "matched", "notmatched", "dbmatched", "dbnotmatched"]),
    "nt_nohit": ("hard_acc_s", ["-userfields", NOHIT_FIELDS, "-output_no_hits"], ["user", "b6"]),
    "nt_top":   ("hard_acc_s", ["-userfields", "query+target+id", "-top_hits_only"], ["user", "uc"]),
    "nt_max2":  ("hard_acc_s", ["-userfields", "query+target+id", "-maxhits", "2"], ["user", "dbmatched"]),
    "nt_top1":  ("hard_acc_s", ["-userfields", "query+target+id", "-top_hit_only"], ["user"]),
    "aa_all":   ("hard_aa_s", ["-userfields", FIELDS], ["user", "b6", "notmatched"]),
    "nt_aln":   ("hard_acc_s", [], ["aln"]),
    "nt_segs":  ("hard_acc_s", [], ["pairs", "qseg", "tseg", "trim"]),
    "aa_aln":   ("hard_aa_s", [], ["aln"]),
}
OPT = {"trim": "-trimout", "pairs": "-fastapairs", "qseg": "-qsegout", "tseg": "-tsegout", "aln": "-alnout", "user": "-userout", "b6": "-blast6out", "uc": "-uc", "matched": "-matched", "notmatched": "-notmatched",
       "dbmatched": "-dbmatched", "dbnotmatched": "-dbnotmatched"}


def main():
    assert os.path.exists(mg.REF)
    manifest = {}
    with tempfile.TemporaryDirectory() as tmp:
        for run, (case, extra, kinds) in RUNS.items():
            c = mg.CASES[case]
            db, qs = mg.make_inputs(c)
            dbfa, qfa = os.path.join(tmp, "db.fa"), os.path.join(tmp, "q.fa")
            db.write_fasta(dbfa)
            qs.write_fasta(qfa)
            cmd = [mg.REF, "-usearch_global", qfa, "-db", dbfa, "-id", str(c["id"]), "-threads", "1"]
            if not c["aa"]:
                cmd += ["-strand", c["strand"]]
            for opt in ("big", "maxaccepts", "maxrejects"):
                if opt in c:
                    cmd += ["-" + opt, str(c[opt])]
            cmd += extra
            for k in kinds:
                cmd += [OPT[k], os.path.join(HERE, "out_%s.%s" % (run, k))]
            subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            files = {}
            for k in kinds:
                path = os.path.join(HERE, "out_%s.%s" % (run, k))
                data = open(path, "rb").read()
                if k == "aln":      # the reference starts -alnout with its command line and a version/RAM banner: not part of the format
                    data = b"".join(data.splitlines(True)[2:])
                    open(path, "wb").write(data)
                files[k] = dict(sha256=hashlib.sha256(data).hexdigest(), lines=data.count(b"\n"), bytes=len(data), whole=len(data) <= 100000)
                if not files[k]["whole"]:
                    os.remove(path)
                    open(path + ".head", "wb").write(b"".join(data.splitlines(True)[:20]))
            manifest[run] = dict(case=case, extra=extra, files=files)
            print(run, {k: v["bytes"] for k, v in files.items()})
    json.dump(manifest, open(os.path.join(HERE, "out_manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
