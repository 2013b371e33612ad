#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the compiled, UNMODIFIED
reference (oracle/_ref/usearch12, built by oracle/build_ref.sh from /root/reference/src)
on seeded synthetic inputs.  Runs only where /root/reference exists (the build container).

Committed per case: <case>.b6 and <case>.uc (the reference's -blast6out / -uc text, -threads 1
so lines are in query order) and one manifest.json with the generator arguments, the
reference command line and sha256 digests of the generated inputs.  Inputs themselves are
NOT committed: tests regenerate them from usearch12_amd/synth.py and check the digests.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from usearch12_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "usearch12")

# name -> dict(gen=..., args=..., id=..., strand=..., opts={reference option: value})
CASES = {
    "nt_small":    dict(gen="uniform", seed=11, db_n=2000, q_n=300, length=250, aa=False, id=0.97, strand="plus"),
    "nt_big":      dict(gen="uniform", seed=12, db_n=3000, q_n=300, length=250, aa=False, id=0.97, strand="plus", big=100),
    "nt_both":     dict(gen="uniform", seed=13, db_n=2000, q_n=300, length=250, aa=False, id=0.97, strand="both"),
    "nt_bigboth":  dict(gen="uniform", seed=14, db_n=3000, q_n=300, length=250, aa=False, id=0.97, strand="both", big=100),
    "aa_small":    dict(gen="uniform", seed=15, db_n=3000, q_n=300, length=300, aa=True, id=0.8),
    "aa_big":      dict(gen="uniform", seed=16, db_n=3000, q_n=300, length=300, aa=True, id=0.8, big=100),
    "nt_lowid":    dict(gen="uniform", seed=17, db_n=2000, q_n=300, length=250, aa=False, id=0.8, strand="plus", mut=[0.08, 0.01, 0.01]),
    "nt_lowidbig": dict(gen="uniform", seed=18, db_n=2000, q_n=300, length=250, aa=False, id=0.8, strand="plus", mut=[0.08, 0.01, 0.01], big=100),
    "nt_short":    dict(gen="uniform", seed=19, db_n=2000, q_n=300, length=60, aa=False, id=0.9, strand="plus", mut=[0.03, 0.01, 0.01]),
    "nt_big120k":  dict(gen="uniform", seed=20, db_n=120000, q_n=400, length=250, aa=False, id=0.97, strand="plus"),
    "hard_small":  dict(gen="hard", seed=21, n_fam=300, fam=8, q_n=1200, aa=False, id=0.97, strand="plus"),
    "hard_big":    dict(gen="hard", seed=22, n_fam=300, fam=8, q_n=1200, aa=False, id=0.97, strand="plus", big=100),
    "hard_both":   dict(gen="hard", seed=23, n_fam=300, fam=8, q_n=1200, aa=False, id=0.97, strand="both", big=100),
    "hard_id90":   dict(gen="hard", seed=24, n_fam=300, fam=8, q_n=1200, aa=False, id=0.90, strand="plus", big=100),
    "hard_id90s":  dict(gen="hard", seed=24, n_fam=300, fam=8, q_n=1200, aa=False, id=0.90, strand="plus"),
    "hard_acc":    dict(gen="hard", seed=25, n_fam=300, fam=8, q_n=800, aa=False, id=0.95, strand="plus", big=100, maxaccepts=4, maxrejects=16),
    "hard_acc_s":  dict(gen="hard", seed=25, n_fam=300, fam=8, q_n=800, aa=False, id=0.95, strand="both", maxaccepts=4, maxrejects=16),
    "hard_aa":     dict(gen="hard", seed=26, n_fam=300, fam=8, q_n=1200, aa=True, id=0.8, big=100),
    "hard_aa_s":   dict(gen="hard", seed=27, n_fam=300, fam=8, q_n=1200, aa=True, id=0.9),
    # mixed lengths 20..300: short queries need wider counters than the batch-typical ones
    "hard_mixlen": dict(gen="hard", seed=28, n_fam=400, fam=6, q_n=1500, aa=False, id=0.97, strand="plus", big=100, lmin=20, lmax=300),
    "hard_mixlen_s": dict(gen="hard", seed=29, n_fam=400, fam=6, q_n=1500, aa=False, id=0.95, strand="both", lmin=20, lmax=300),
    # optional accept filters of Accepter::IsAcceptLo (a failed filter is a reject for the terminator)
    "hard_filt":   dict(gen="hard", seed=30, n_fam=300, fam=8, q_n=1200, aa=False, id=0.90, strand="plus", big=100, maxaccepts=4, maxrejects=16,
                        query_cov=0.9, target_cov=0.85, maxgaps=3, maxdiffs=25, mindiffs=1, maxid=0.995, mincols=150),
    "hard_filt_s": dict(gen="hard", seed=31, n_fam=300, fam=8, q_n=1200, aa=False, id=0.92, strand="both", maxaccepts=3, maxrejects=12,
                        max_query_cov=0.999, max_target_cov=0.995, maxgaps=6, maxdiffs=30, lmin=100, lmax=400),
    # -fulldp (one unbanded Viterbi per pair, no HSPs) and -gaforce (pairs without good HSPs are aligned all the same)
    "hard_fulldp":  dict(gen="hard", seed=33, n_fam=250, fam=6, q_n=900, aa=False, id=0.9, strand="both", lmin=20, lmax=300, maxaccepts=2, maxrejects=8, fulldp=1),
    "hard_gaforce": dict(gen="hard", seed=34, n_fam=250, fam=6, q_n=900, aa=False, id=0.85, strand="plus", big=100, lmin=20, lmax=300, gaforce=1),
    "hard_fulldp_aa": dict(gen="hard", seed=35, n_fam=200, fam=6, q_n=600, aa=True, id=0.7, big=100, lmin=30, lmax=250, fulldp=1, gaforce=1),
    "hard_hardmask": dict(gen="hard", seed=36, n_fam=250, fam=6, q_n=900, aa=False, id=0.9, strand="both", big=100, maxaccepts=2, maxrejects=8, hardmask=1),
    "hard_hardmask_aa": dict(gen="hard", seed=37, n_fam=250, fam=6, q_n=700, aa=True, id=0.7, maxaccepts=2, maxrejects=8, hardmask=1),
    # -termid / -termidd: the walk also ends on the identity of the hits collected so far (both strands share them)
    "hard_termid":  dict(gen="hard", seed=38, n_fam=250, fam=8, q_n=900, aa=False, id=0.8, strand="both", big=100, maxaccepts=6, maxrejects=16, termid=0.93),
    "hard_termidd": dict(gen="hard", seed=39, n_fam=250, fam=8, q_n=900, aa=False, id=0.8, strand="both", maxaccepts=6, maxrejects=16, termidd=0.04),
    # -band 0: every hole (and a pair without HSPs under -gaforce) goes through the unbanded ViterbiFastMem (globalalignmem.cpp:105-108,118-119)
    "hard_band0":  dict(gen="hard", seed=40, n_fam=250, fam=6, q_n=900, aa=False, id=0.9, strand="both", big=100, lmin=20, lmax=300, maxaccepts=2, maxrejects=8, band=0, gaforce=1),
    "hard_band0_aa": dict(gen="hard", seed=41, n_fam=200, fam=6, q_n=600, aa=True, id=0.7, lmin=30, lmax=250, band=0),
    # maxaccepts 0 / maxrejects 0 = unlimited (terminator.cpp:40-45,91-97): the walk ends on the other limit or at the end of the list
    "hard_acc0":   dict(gen="hard", seed=42, n_fam=250, fam=6, q_n=900, aa=False, id=0.9, strand="both", big=100, lmin=20, lmax=300, maxaccepts=0, maxrejects=6),
    "hard_rej0":   dict(gen="hard", seed=43, n_fam=7, fam=8, q_n=600, aa=False, id=0.9, strand="plus", lmin=100, lmax=300, maxaccepts=2, maxrejects=0),
    # no -id at all: ranking parameters of id 0.5, no identity filter, no error (accepter.cpp:35, makedbsearcher.cpp:195)
    "hard_noid":   dict(gen="hard", seed=44, n_fam=250, fam=6, q_n=900, aa=False, id=None, strand="both", big=100, lmin=20, lmax=300, maxaccepts=3, maxrejects=8),
    "hard_noid_s": dict(gen="hard", seed=45, n_fam=200, fam=6, q_n=700, aa=False, id=None, strand="plus", lmin=60, lmax=300),
    # deep walks (r5): more candidates than the 64 a ranking pass keeps - unlimited walks over a DB of large families (every family member is a
    # candidate, up to ~100 accepts per query: hit slots overflow into chained blocks), -maxrejects 128 / 256, -maxaccepts 100; both ranking paths,
    # both strands, protein.  The reference's walk has no depth limit (terminator.cpp:22-31,64-100).
    "deep_all_s":   dict(gen="hard", seed=46, n_fam=25, fam=120, q_n=100, aa=False, id=0.9, strand="both", lmin=100, lmax=300, maxaccepts=0, maxrejects=0),
    "deep_all_big": dict(gen="hard", seed=47, n_fam=25, fam=120, q_n=100, aa=False, id=0.9, strand="plus", big=100, lmin=100, lmax=300, maxaccepts=0, maxrejects=0),
    "deep_rej256":  dict(gen="hard", seed=48, n_fam=25, fam=120, q_n=300, aa=False, id=0.97, strand="both", big=100, lmin=100, lmax=300, maxaccepts=2, maxrejects=256),
    "deep_rej128_s": dict(gen="hard", seed=49, n_fam=25, fam=120, q_n=300, aa=False, id=0.95, strand="plus", lmin=100, lmax=300, maxaccepts=3, maxrejects=128),
    "deep_acc100":  dict(gen="hard", seed=50, n_fam=25, fam=120, q_n=150, aa=False, id=0.9, strand="plus", big=100, lmin=100, lmax=300, maxaccepts=100, maxrejects=32),
    "deep_aa":      dict(gen="hard", seed=51, n_fam=20, fam=100, q_n=100, aa=True, id=0.8, lmin=80, lmax=250, maxaccepts=0, maxrejects=0),
    "deep_aa_big":  dict(gen="hard", seed=52, n_fam=20, fam=100, q_n=120, aa=True, id=0.85, big=100, lmin=80, lmax=250, maxaccepts=1, maxrejects=200),
    # options of the index and of the aligner that the other cases leave at their defaults (r5): -wordlength (udbparams.cpp), -stepwords / -bump
    # (small-path sampling, udbusortedsearcher.cpp), -minhsp / -xdrop_nw / -hspw (hspfinder.cpp), -match / -mismatch (alnparams.cpp)
    "opt_word6":    dict(gen="hard", seed=60, n_fam=250, fam=6, q_n=800, aa=False, id=0.9, strand="both", big=100, lmin=40, lmax=300, wordlength=6),
    "opt_word7_s":  dict(gen="hard", seed=61, n_fam=250, fam=6, q_n=800, aa=False, id=0.95, strand="plus", lmin=40, lmax=300, wordlength=7),
    "opt_word4_aa": dict(gen="hard", seed=62, n_fam=200, fam=6, q_n=600, aa=True, id=0.7, big=100, lmin=30, lmax=250, wordlength=4),
    "opt_step0_s":  dict(gen="hard", seed=63, n_fam=250, fam=6, q_n=800, aa=False, id=0.9, strand="plus", lmin=60, lmax=300, stepwords=0),
    "opt_step3_s":  dict(gen="hard", seed=64, n_fam=250, fam=6, q_n=800, aa=False, id=0.9, strand="both", lmin=60, lmax=300, stepwords=3, bump=10),
    "opt_bump90_s": dict(gen="hard", seed=65, n_fam=250, fam=6, q_n=800, aa=False, id=0.95, strand="plus", lmin=60, lmax=300, bump=90),
    "opt_minhsp":   dict(gen="hard", seed=66, n_fam=250, fam=6, q_n=800, aa=False, id=0.85, strand="both", big=100, lmin=20, lmax=300, minhsp=24, xdrop_nw=4),
    "opt_minhsp8":  dict(gen="hard", seed=67, n_fam=250, fam=6, q_n=800, aa=False, id=0.85, strand="plus", lmin=20, lmax=300, minhsp=8, xdrop_nw=16),
    "opt_match":    dict(gen="hard", seed=68, n_fam=250, fam=6, q_n=800, aa=False, id=0.85, strand="plus", big=100, lmin=20, lmax=300, match=2.0, mismatch=-3.0),
    "opt_hspw4":    dict(gen="hard", seed=69, n_fam=250, fam=6, q_n=800, aa=False, id=0.85, strand="both", big=100, lmin=20, lmax=300, hspw=4),
    "opt_hspw2_aa": dict(gen="hard", seed=70, n_fam=200, fam=6, q_n=600, aa=True, id=0.7, lmin=30, lmax=250, hspw=2, minhsp=10),
    # -dbmask none (everything upper case: the lower-case stretches of the hard set become ordinary letters) / user (letters as given: lower case
    # voids words and never counts as identical), makeudb.cpp:11-25, seqdb.cpp:415-449
    "opt_mask_none": dict(gen="hard", seed=71, n_fam=250, fam=6, q_n=800, aa=False, id=0.9, strand="both", big=100, lmin=40, lmax=300, dbmask="none"),
    "opt_mask_user": dict(gen="hard", seed=72, n_fam=250, fam=6, q_n=800, aa=False, id=0.9, strand="plus", lmin=40, lmax=300, dbmask="user"),
    "opt_mask_none_aa": dict(gen="hard", seed=73, n_fam=200, fam=6, q_n=600, aa=True, id=0.7, big=100, lmin=30, lmax=250, dbmask="none"),
    "opt_mask_user_aa": dict(gen="hard", seed=74, n_fam=200, fam=6, q_n=600, aa=True, id=0.7, lmin=30, lmax=250, dbmask="user"),
    "hard_filt_aa": dict(gen="hard", seed=32, n_fam=300, fam=8, q_n=1000, aa=True, id=0.8, big=100, maxaccepts=2, maxrejects=16,
                         query_cov=0.95, maxgaps=4, mindiffs=3),
}
EXTRA_OPTS = ("wordlength", "stepwords", "bump", "minhsp", "xdrop_nw", "match", "mismatch", "hspw", "dbmask")      # reference option -> ugs_params field: golden_util.params_kw
FILTER_OPTS = ("maxid", "mincols", "maxgaps", "query_cov", "max_query_cov", "target_cov", "max_target_cov", "maxdiffs", "mindiffs")


def make_inputs(c):
    if c["gen"] == "uniform":
        db = synth.make_db(c["seed"], c["db_n"], c["length"], c["aa"])
        kw = {}
        if "mut" in c:
            kw = dict(p_sub=c["mut"][0], p_del=c["mut"][1], p_ins=c["mut"][2])
        qs = synth.make_queries(c["seed"], db, c["q_n"], c["length"], c["aa"], **kw)
    else:
        db, qs = synth.make_hard(c["seed"], c["n_fam"], c["fam"], c["q_n"], lmin=c.get("lmin", 150), lmax=c.get("lmax", 400),
                                 aa=c["aa"])
    if c.get("strand") == "both":
        qs = synth.revcomp_some(c["seed"], qs)
    return db, qs


def digest(ss):
    h = hashlib.sha256()
    h.update(ss.offs.tobytes())
    h.update(ss.seqs.tobytes())
    return h.hexdigest()


def ref_cmd(c, qfa, dbfa, prefix):
    cmd = [REF, "-usearch_global", qfa, "-db", dbfa] + (["-id", str(c["id"])] if c["id"] is not None else []) + \
          ["-blast6out", prefix + ".b6", "-uc", prefix + ".uc", "-threads", "1"]
    if not c["aa"]:
        cmd += ["-strand", c["strand"]]
    for opt in ("big", "maxaccepts", "maxrejects", "band") + FILTER_OPTS:
        if opt in c:
            cmd += ["-" + opt, str(c[opt])]
    for opt in ("termid", "termidd") + EXTRA_OPTS:
        if opt in c:
            cmd += ["-" + opt, str(c[opt])]
    for flag in ("fulldp", "gaforce", "hardmask"):
        if c.get(flag):
            cmd += ["-" + flag]
    return cmd


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
            subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            nb6 = sum(1 for _ in open(prefix + ".b6"))
            manifest[name] = dict(c, db_sha256=digest(db), q_sha256=digest(qs), n_hits=nb6,
                                  cmd=" ".join(["usearch12"] + [os.path.basename(x) if x.startswith(tmp) else x
                                                                 for x in cmd[1:]]).replace(HERE + "/", ""))
            print(name, "hits", nb6)
    path = os.path.join(HERE, "manifest.json")
    if only and os.path.exists(path):
        old = json.load(open(path))
        old.update(manifest)
        manifest = old
    json.dump(manifest, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
