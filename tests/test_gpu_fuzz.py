"""Randomised differential test: the HIP path (through the C-ABI) against the oracle on seeded random
configurations - small databases of sequence families with low-complexity runs, wildcards and lower-case stretches,
odd lengths (down to shorter than a word), duplicated sequences, random option mixes (identity, accepts/rejects,
both strands, small/Big ranker, -stepwords, -bump, accept filters).  Every hit record and path must be identical."""
import os

import numpy as np
import pytest

import orc
from usearch12_amd import capi, synth

pytestmark = pytest.mark.gpu


def config(seed):
    rng = np.random.default_rng([seed, 0xF022])
    aa = bool(rng.random() < 0.3)
    big_shape = os.environ.get("UGS_FUZZ_BIG") is not None          # ad-hoc sweeps: longer sequences, larger databases
    lmin = int(rng.choice([4, 12, 30, 60, 150] + ([600, 1500] if big_shape else [])))
    lmax = lmin + int(rng.choice([0, 20, 100, 300]))
    n_fam, fam = int(rng.integers(5, 2000 if big_shape else 200)), int(rng.integers(1, 12))
    nq = int(rng.integers(50, 600))
    kw = dict(max_accepts=int(rng.integers(1, 5)), max_rejects=int(rng.choice([1, 4, 8, 16, 32])))
    if not aa and rng.random() < 0.5:
        kw["strand_both"] = 1
    if rng.random() < 0.6:
        kw["big"] = int(rng.choice([1, 50, 400]))             # force the Big ranker on small databases
    if rng.random() < 0.3:
        kw["stepwords"] = int(rng.choice([0, 1, 3, 20]))
    if rng.random() < 0.2:
        kw["bump_pct"] = int(rng.choice([0, 10, 90]))
    ident = float(rng.choice([0.5, 0.7, 0.8, 0.9, 0.97, 0.99, 1.0])) if not aa else float(rng.choice([0.5, 0.6, 0.8, 0.95]))
    if rng.random() < 0.35:
        opts = dict(maxid=0.995, mincols=int(lmin * 0.8), maxgaps=int(rng.integers(0, 6)), query_cov=0.8, max_query_cov=0.99,
                    target_cov=0.7, max_target_cov=0.98, maxdiffs=int(rng.integers(1, 30)), mindiffs=int(rng.integers(1, 4)))
        for k in rng.choice(sorted(opts), size=int(rng.integers(1, 4)), replace=False):
            kw[str(k)] = opts[str(k)]
    if seed >= 48:
        # r5, deep walks: walk depths beyond the 64 candidates of a ranking pass (0 = unlimited) over databases of LARGE families, so that
        # lists are hundreds of candidates long and queries collect more hits than a unit's slots
        r2 = np.random.default_rng([seed, 0xDEE9])
        kw["max_accepts"] = int(r2.choice([0, 1, 3, 70, 100]))
        kw["max_rejects"] = int(r2.choice([0, 40, 100, 200]))
        if kw["max_accepts"] and kw["max_rejects"] and kw["max_accepts"] + kw["max_rejects"] - 1 <= 64:
            kw["max_rejects"] = 100
        n_fam, fam = int(r2.integers(3, 15)), int(r2.integers(40, 130))
        nq = int(r2.integers(30, 150))
        lmin = max(lmin, 30)
        lmax = max(lmax, lmin)
    return aa, lmin, lmax, n_fam, fam, nq, ident, kw


@pytest.mark.parametrize("seed", range(int(os.environ.get("UGS_FUZZ_FROM", 0)), int(os.environ.get("UGS_FUZZ_TO", 64))))
def test_random_configuration_matches_oracle(seed):
    aa, lmin, lmax, n_fam, fam, nq, ident, kw = config(seed)
    db, qs = synth.make_hard(1000 + seed, n_fam, fam, nq, lmin=lmin, lmax=lmax, aa=aa)
    if kw.get("strand_both"):
        qs = synth.revcomp_some(seed, qs)
    if seed % 5 == 0:                                           # exact duplicates in the database (ties everywhere)
        seqs = [np.frombuffer(db.seq(i), np.uint8) for i in list(range(db.n)) + list(range(0, db.n, 3))]
        offs = np.zeros(len(seqs) + 1, np.uint64)
        offs[1:] = np.cumsum([len(x) for x in seqs])
        db = synth.SeqSet(np.concatenate(seqs), offs, lambda i: "d%d" % i)
    hits, nh, pool = capi.UgsDB(capi.params(is_nucleo=not aa, id=ident, **kw), db.seqs, db.offs, device=0).search(qs.seqs, qs.offs)
    oh, onh, opool = orc.OrcDB(orc.params(is_nucleo=not aa, id=ident, **kw), db.seqs, db.offs).search(qs.seqs, qs.offs, nthreads=4)
    assert np.array_equal(nh, onh), (seed, kw)
    for f in hits.dtype.names:
        if f != "cigar_off":
            assert np.array_equal(hits[f], oh[f]), (seed, f, kw)
    for a, b in zip(hits, oh):
        assert np.array_equal(pool[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["cigar_len"])],
                              opool[int(b["cigar_off"]):int(b["cigar_off"]) + int(b["cigar_len"])]), (seed, kw)


def local_config(seed):
    rng = np.random.default_rng([seed, 0x10CA1])
    aa = bool(rng.random() < 0.35)
    lmin = int(rng.choice([8, 30, 60, 150, 400]))
    lmax = lmin + int(rng.choice([0, 40, 200, 600]))
    n_fam, fam = int(rng.integers(5, 150)), int(rng.integers(1, 8))
    nq = int(rng.integers(50, 400))
    kw = dict(max_accepts=int(rng.integers(1, 5)), max_rejects=int(rng.choice([1, 4, 8, 16])), max_hsps=32,
              local_evalue=float(rng.choice([10.0, 1.0, 1e-3, 1e-6, 1e-20])), id=None)
    if not aa and rng.random() < 0.5:
        kw["strand_both"] = 1
    if rng.random() < 0.5:
        kw["id"] = float(rng.choice([0.5, 0.7, 0.9]))
        if rng.random() < 0.6:
            kw["big"] = int(rng.choice([1, 50]))                # the Big ranker needs -id in the reference
    if rng.random() < 0.4:
        kw["xdrop_u"] = float(rng.choice([4.0, 9.5, 16.0, 40.0]))
        kw["xdrop_g"] = float(rng.choice([6.0, 20.0, 32.0, 64.0]))
    if rng.random() < 0.3:
        kw["hsp_word_len"] = int(rng.choice([3, 4, 6] if not aa else [2, 3]))
    if rng.random() < 0.3:
        opts = dict(mincols=int(lmin * 0.5), maxgaps=int(rng.integers(0, 6)), query_cov=0.3, target_cov=0.2, maxdiffs=int(rng.integers(1, 40)))
        for k in rng.choice(sorted(opts), size=int(rng.integers(1, 3)), replace=False):
            kw[str(k)] = opts[str(k)]
    return aa, lmin, lmax, n_fam, fam, nq, kw


def _same(hits, nh, pool, oh, onh, opool, ctx):
    assert np.array_equal(nh, onh), ctx
    for f in hits.dtype.names:
        if f != "cigar_off":
            assert np.array_equal(hits[f], oh[f]), (f,) + ctx
    for a, b in zip(hits, oh):
        assert np.array_equal(pool[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["cigar_len"])],
                              opool[int(b["cigar_off"]):int(b["cigar_off"]) + int(b["cigar_len"])]), ctx


@pytest.mark.parametrize("seed", range(int(os.environ.get("UGS_LFUZZ_FROM", 0)), int(os.environ.get("UGS_LFUZZ_TO", 24))))
def test_random_local_configuration_matches_oracle(seed):
    aa, lmin, lmax, n_fam, fam, nq, kw = local_config(seed)
    db, _ = synth.make_hard(3000 + seed, n_fam, fam, 1, lmin=lmin, lmax=lmax, aa=aa)
    qs = synth.make_local_queries(3000 + seed, db, nq, aa=aa)
    if kw.get("strand_both"):
        qs = synth.revcomp_some(seed, qs)
    got = capi.UgsDB(capi.params(is_nucleo=not aa, **kw), db.seqs, db.offs, device=0).search(qs.seqs, qs.offs)
    want = orc.OrcDB(orc.params(is_nucleo=not aa, **kw), db.seqs, db.offs).search(qs.seqs, qs.offs, nthreads=4)
    _same(*got, *want, (seed, kw))


def extras_config(seed):
    rng = np.random.default_rng([seed, 0xE17A])
    lmin = int(rng.choice([12, 60, 150]))
    lmax = lmin + int(rng.choice([0, 50, 250]))
    kw = dict(max_accepts=int(rng.integers(1, 4)), max_rejects=int(rng.choice([2, 8, 16])))
    if rng.random() < 0.5:
        kw["strand_both"] = 1
    if rng.random() < 0.5:
        kw["big"] = int(rng.choice([1, 60]))
    r = rng.random()
    if r < 0.25:
        kw["align_flags"] = int(rng.choice([1, 2, 3]))
    elif r < 0.5:
        kw[str(rng.choice(["self", "selfid"]))] = True
    elif r < 0.75:
        for k, v in (("minqt", 0.8), ("maxqt", 1.2), ("minsl", 0.7), ("maxsl", 0.98)):
            if rng.random() < 0.5:
                kw[k] = v
        if not any(k in kw for k in ("minqt", "maxqt", "minsl", "maxsl")):
            kw["minsl"] = 0.8
    else:
        kw[str(rng.choice(["min_sizeratio", "abskew"]))] = float(rng.choice([0.5, 1.0, 3.0]))
    return lmin, lmax, int(rng.integers(5, 150)), int(rng.integers(1, 8)), int(rng.integers(50, 400)), float(rng.choice([0.7, 0.9, 0.97])), kw


@pytest.mark.parametrize("seed", range(int(os.environ.get("UGS_XFUZZ_FROM", 0)), int(os.environ.get("UGS_XFUZZ_TO", 24))))
def test_random_pair_filter_and_aligner_options_match_oracle(seed):
    lmin, lmax, n_fam, fam, nq, ident, kw = extras_config(seed)
    db, qs = synth.make_hard(5000 + seed, n_fam, fam, nq, lmin=lmin, lmax=lmax, aa=False)
    if kw.get("self") or kw.get("selfid"):                   # all-vs-all with some exact duplicates
        seqs = [np.frombuffer(db.seq(i), np.uint8) for i in list(range(db.n)) + list(range(0, db.n, 4))]
        offs = np.zeros(len(seqs) + 1, np.uint64)
        offs[1:] = np.cumsum([len(x) for x in seqs])
        db = synth.SeqSet(np.concatenate(seqs), offs, lambda i: "s%d" % (i if i < len(seqs) - len(range(0, db.n, 4)) else -i))
        qs = db
    elif kw.get("strand_both"):
        qs = synth.revcomp_some(seed, qs)
    rng = np.random.default_rng(seed)
    tk = np.arange(db.n, dtype=np.uint32)
    qk = tk if qs is db else np.arange(qs.n, dtype=np.uint32) + np.uint32(db.n)
    tz = rng.integers(1, 50, db.n).astype(np.uint32)
    qz = tz if qs is db else rng.integers(1, 50, qs.n).astype(np.uint32)
    gdb = capi.UgsDB(capi.params(is_nucleo=True, id=ident, **kw), db.seqs, db.offs, device=0)
    gdb.set_pair_keys(tk, tz)
    got = gdb.search(qs.seqs, qs.offs, pair_keys=(qk, qz))
    odb = orc.OrcDB(orc.params(is_nucleo=True, id=ident, **kw), db.seqs, db.offs)
    odb.set_pair_keys(tk, tz); odb.set_query_pair_keys(qk, qz)
    want = odb.search(qs.seqs, qs.offs, nthreads=4)
    _same(*got, *want, (seed, kw))
