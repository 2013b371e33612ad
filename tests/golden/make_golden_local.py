#!/usr/bin/env python3
"""Known answers for the usearch_local row, produced by the UNMODIFIED reference
(oracle/_ref/usearch12 -usearch_local ...) on seeded synthetic inputs.  Runs only where /root/reference
exists.  Committed per case: <case>.b6 (-blast6out, -threads 1 => query order) and local_manifest.json with
the generator arguments, the reference command line and sha256 digests of the regenerated inputs."""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from usearch12_amd import synth  # noqa: E402
from make_golden import digest, REF, FILTER_OPTS  # noqa: E402

CASES = {
    "loc_nt":       dict(seed=41, n_fam=250, fam=6, q_n=900, aa=False, evalue=1e-6, strand="plus"),
    "loc_nt_both":  dict(seed=42, n_fam=250, fam=6, q_n=900, aa=False, evalue=1e-3, strand="both", maxaccepts=3, maxrejects=8),
    "loc_nt_id":    dict(seed=43, n_fam=250, fam=6, q_n=900, aa=False, evalue=1e-9, strand="both", id=0.9, maxaccepts=2, maxrejects=6,
                         mincols=60, maxgaps=8, query_cov=0.3),
    "loc_nt_big":   dict(seed=44, n_fam=250, fam=6, q_n=600, aa=False, evalue=1e-6, strand="plus", id=0.8, big=100),
    "loc_aa":       dict(seed=45, n_fam=250, fam=6, q_n=900, aa=True, evalue=1e-6),
    "loc_aa_acc":   dict(seed=46, n_fam=250, fam=6, q_n=900, aa=True, evalue=10.0, maxaccepts=4, maxrejects=4, target_cov=0.2, maxdiffs=60),
    "loc_nt_long":  dict(seed=47, n_fam=40, fam=4, q_n=120, aa=False, evalue=1e-6, strand="both", lmin=800, lmax=3000),
    # deep walks (r5): more candidates than the 64 a ranking pass keeps - families of 100, unlimited walks / -maxrejects 128 / -maxaccepts 80
    "loc_deep_nt":  dict(seed=48, n_fam=20, fam=100, q_n=80, aa=False, evalue=1e-6, strand="both", maxaccepts=0, maxrejects=0),
    "loc_deep_aa":  dict(seed=49, n_fam=20, fam=100, q_n=120, aa=True, evalue=1e-3, maxaccepts=80, maxrejects=128),
    # options every other case leaves at their defaults (r5): -xdrop_u / -xdrop_g / -maxhsps / -hspw (localaligner2.cpp), -ka_dbsize (estats.cpp),
    # -lopen / -lext (alnparams.cpp), -match / -mismatch
    "loc_opt_nt":   dict(seed=51, n_fam=250, fam=6, q_n=700, aa=False, evalue=1e-6, strand="both", xdrop_u=8, xdrop_g=16, hspw=4),
    "loc_opt_aa":   dict(seed=52, n_fam=250, fam=6, q_n=700, aa=True, evalue=1e-3, xdrop_u=24, xdrop_g=48, ka_dbsize=1e6, hspw=2),
    "loc_opt_gap":  dict(seed=53, n_fam=250, fam=6, q_n=700, aa=False, evalue=1e-6, strand="plus", lopen=6.0, lext=2.0, maxaccepts=2, maxrejects=8),
    "loc_opt_gap_aa": dict(seed=54, n_fam=250, fam=6, q_n=700, aa=True, evalue=1e-6, lopen=14.0, lext=0.5),
    "loc_opt_match": dict(seed=55, n_fam=250, fam=6, q_n=700, aa=False, evalue=1e-6, strand="both", match=2.0, mismatch=-3.0, id=0.8, big=100),
    "loc_deep_big": dict(seed=50, n_fam=20, fam=100, q_n=100, aa=False, evalue=1e-6, strand="plus", id=0.8, big=100, maxaccepts=2, maxrejects=200),
}


def make_inputs(c):
    db, _ = synth.make_hard(c["seed"], c["n_fam"], c["fam"], 1, lmin=c.get("lmin", 150), lmax=c.get("lmax", 400), aa=c["aa"])
    qs = synth.make_local_queries(c["seed"], db, c["q_n"], aa=c["aa"])
    if c.get("strand") == "both":
        qs = synth.revcomp_some(c["seed"], qs)
    return db, qs


def ref_cmd(c, qfa, dbfa, prefix):
    cmd = [REF, "-usearch_local", qfa, "-db", dbfa, "-evalue", repr(c["evalue"]), "-blast6out", prefix + ".b6", "-threads", "1"]
    if not c["aa"]:
        cmd += ["-strand", c["strand"]]
    for opt in ("id", "big", "maxaccepts", "maxrejects") + FILTER_OPTS + LOCAL_OPTS:
        if opt in c:
            cmd += ["-" + opt, str(c[opt])]
    return cmd


LOCAL_OPTS = ("xdrop_u", "xdrop_g", "maxhsps", "ka_dbsize", "hspw", "lopen", "lext", "match", "mismatch")      # -> ugs_params: golden_util.local_params_kw


USER_FIELDS = ("query+target+clusternr+evalue+id+fractid+dist+mid+pctpv+pctgaps+pairs+gaps+allgaps+qlo+qhi+tlo+thi+qlor+qhir+tlor+thir+"
               "qlot+qhit+qunt+tlot+thit+tunt+pv+ql+tl+qs+ts+alnlen+opens+exts+raw+bits+aln+caln+qseq+tseq+qseg+tseg+qstrand+tstrand+qrow+trow+"
               "qrowdots+trowdots+qframe+tframe+orflo+orfhi+orfframe+mism+ids+qcov+tcov+diffs+diffsa+editdiffs")
USER_CASES = ("loc_nt_both", "loc_aa_acc")      # -userout with every supported field: kept as sha256 + first 12 lines


def main():
    assert os.path.exists(REF), "build the reference first: oracle/build_ref.sh"
    manifest = {}
    only = set(sys.argv[1:])
    with tempfile.TemporaryDirectory() as tmp:
        for name, c in CASES.items():
            if only and name not in only:
                continue
            db, qs = make_inputs(c)
            dbfa, qfa = os.path.join(tmp, "db.fa"), os.path.join(tmp, "q.fa")
            db.write_fasta(dbfa)
            qs.write_fasta(qfa)
            prefix = os.path.join(HERE, name)
            cmd = ref_cmd(c, qfa, dbfa, prefix)
            if name in USER_CASES:
                cmd += ["-userout", os.path.join(tmp, "u.txt"), "-userfields", USER_FIELDS, "-alnout", os.path.join(tmp, "a.txt"), "-uc", os.path.join(tmp, "c.uc")]
            subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            user = None
            if name in USER_CASES:
                import hashlib
                data = open(os.path.join(tmp, "u.txt"), "rb").read()
                user = dict(sha256=hashlib.sha256(data).hexdigest(), lines=data.count(b"\n"), fields=USER_FIELDS)
                open(prefix + ".user.head", "wb").write(b"".join(data.splitlines(True)[:12]))
                aln = b"".join(open(os.path.join(tmp, "a.txt"), "rb").read().splitlines(True)[2:])     # minus the command-line / version banner
                user["aln_sha256"] = hashlib.sha256(aln).hexdigest(); user["aln_lines"] = aln.count(b"\n")
                open(prefix + ".aln.head", "wb").write(b"".join(aln.splitlines(True)[:60]))
                ucd = open(os.path.join(tmp, "c.uc"), "rb").read()
                user["uc_sha256"] = hashlib.sha256(ucd).hexdigest(); user["uc_lines"] = ucd.count(b"\n")
                cmd = cmd[:-8]
                # hit-count rules on local scores: -top_hits_only, -top_hit_only (HitMgr::GetHitCount / GetTopHit)
                for flag in ("top_hits_only", "top_hit_only"):
                    subprocess.check_call(cmd[:cmd.index("-blast6out")] + ["-blast6out", os.path.join(tmp, flag + ".b6"), "-" + flag] + cmd[cmd.index("-blast6out") + 2:],
                                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                    d = open(os.path.join(tmp, flag + ".b6"), "rb").read()
                    user[flag + "_sha256"] = hashlib.sha256(d).hexdigest(); user[flag + "_lines"] = d.count(b"\n")
            lines = open(prefix + ".b6").read().splitlines()
            pairs = {}
            for ln in lines:
                f = ln.split("\t")
                pairs[(f[0], f[1])] = pairs.get((f[0], f[1]), 0) + 1
            manifest[name] = dict(c, db_sha256=digest(db), q_sha256=digest(qs), n_hits=len(lines), userout=user,
                                  n_multi_hsp_pairs=sum(1 for v in pairs.values() if v > 1),
                                  cmd=" ".join(["usearch12"] + [os.path.basename(x) if x.startswith(tmp) else x
                                                                 for x in cmd[1:]]).replace(HERE + "/", ""))
            print(name, "hits", len(lines), "pairs with >1 HSP", manifest[name]["n_multi_hsp_pairs"])
    path = os.path.join(HERE, "local_manifest.json")
    if only and os.path.exists(path):
        old = json.load(open(path)); old.update(manifest); manifest = old
    json.dump(manifest, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
