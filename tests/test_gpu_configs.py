"""The BASELINE.json configurations at their literal database sizes on one GPU (C1: 10k x 250 nt vs 50k, small ranking path;
one C4 shard: 125k of 10M x 250 nt queries vs the full 5M-sequence DB, i.e. the share of one of 8 GPUs at 1/10 of its length;
C5: 1M x 300 aa vs 2M aa, -id 0.8): the whole query set runs on the GPU, a sample of it through the oracle (its
single-threaded index build over the full database dominates the run time), every hit record and sampled path compared;
plus size-independent properties over the full hit table."""
import os

import numpy as np
import pytest

import orc
from usearch12_amd import capi, synth

pytestmark = pytest.mark.gpu


def _check(db, qs, aa, ident, n_oracle, min_hit_frac):
    p = capi.params(is_nucleo=not aa, id=ident)
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
    bat.upload(qs.seqs, qs.offs)
    bat.search(); bat.sync()
    hits, nh, pool = bat.fetch()
    # properties over the whole table: one accept per query, identities at or above the threshold, sources recovered
    assert nh.max() <= 1 and len(hits) == int(nh.sum())
    assert len(hits) >= min_hit_frac * qs.n
    ident_f = hits["ids"].astype(np.float64) / np.maximum(hits["aln_len"], 1)
    assert ident_f.min() >= float(np.float32(ident)) - 1e-12
    assert np.all(hits["target"] < db.n) and np.all(np.diff(hits["query"].astype(np.int64)) > 0)
    src = qs.src[hits["query"]]
    assert np.mean(src == hits["target"]) > 0.99          # a mutated copy finds the sequence it was made from
    # the oracle on a prefix of the queries
    odb = orc.OrcDB(orc.params(is_nucleo=not aa, id=ident), db.seqs, db.offs)
    so = qs.offs[:n_oracle + 1].copy()
    oh, onh, opool = odb.search(qs.seqs[:int(so[-1])], so, nthreads=min(64, os.cpu_count() or 1))
    k = int(nh[:n_oracle].sum())
    assert np.array_equal(nh[:n_oracle], onh)
    for f in hits.dtype.names:
        if f != "cigar_off":
            assert np.array_equal(hits[:k][f], oh[f]), f
    for a, b in zip(hits[:k][::17], oh[::17]):
        assert np.array_equal(pool[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["cigar_len"])],
                              opool[int(b["cigar_off"]):int(b["cigar_off"]) + int(b["cigar_len"])])


def test_c1_literal_size_small_ranking_path():
    db = synth.make_db(1, 50_000, 250)
    qs = synth.make_queries(1, db, 10_000, 250)
    _check(db, qs, False, 0.97, 2_000, 0.75)


def test_c4_shard_against_the_full_5m_database():
    db = synth.make_db(4, 5_000_000, 250)
    qs = synth.make_queries(4, db, 125_000, 250)
    _check(db, qs, False, 0.97, 1_500, 0.75)


def test_c5_protein_literal_size():
    db = synth.make_db(5, 2_000_000, 300, aa=True)
    qs = synth.make_queries(5, db, 1_000_000, 300, aa=True)
    _check(db, qs, True, 0.8, 1_500, 0.75)


def test_c4_as_named_eight_shards_gathered_to_rank_0():
    """BASELINE.json configs[3] as written, on the one GPU a test box has: the 10 M x 250 nt seed-4 query stream
    (bench.py's generator: the same queries whatever the sharding) in EIGHT contiguous shards of 1.25 M, one rank per shard - each rank
    with its own replica of the 5 M-sequence index (8 x 6 GB of HBM), its own batch object and host thread - and ONE gather of the
    eight device-resident hit tables to rank 0 through the product's C++ gather (ugs_gather_results; loopback transport, because
    RCCL refuses two ranks on one device: every step of the gather except ncclSend / ncclRecv themselves).  The merged table is
    checked by the property set of _check over all 10 M queries and against the oracle on a sample spread over all eight shards,
    both sides of every shard boundary included."""
    import threading
    import bench
    world, total_q, L = 8, 10_000_000, 250
    db = synth.make_db(4, 5_000_000, L)
    p = capi.params(is_nucleo=True, id=0.97)
    comms = capi.UgsComm.init_loopback(world, 0)
    res, err, bounds = [None] * world, [], []
    from usearch12_amd import multigpu
    for r in range(world):
        bounds.append(multigpu.shard_range(total_q, world, r))
    assert bounds[0][0] == 0 and bounds[-1][1] == total_q and all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
    samples = {}           # global query index -> letters (what the oracle searches)
    lock = threading.Lock()
    tiny = capi.UgsDB(p, db.seqs[:int(db.offs[100])], db.offs[:101].copy(), device=0)      # (only for a failing rank's way into the collective)

    def run(r):
        try:
            lo, hi = bounds[r]
            (qs,) = bench.make_query_sets(synth, 4, db, lo, hi - lo, L, 1)
            gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)                 # this rank's replica of the index
            bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
            bat.upload(qs.seqs, qs.offs); bat.search(); bat.sync()
            with lock:
                for i in list(range(0, 150)) + list(range(qs.n - 150, qs.n)):
                    samples[lo + i] = qs.seqs[int(qs.offs[i]):int(qs.offs[i + 1])].copy()
            res[r] = comms[r].gather(bat, lo, dst=0)
            bat.close(); gdb.close()
        except Exception as e:                                                # a failing rank must not leave the others in the barrier
            err.append((r, e))
            try:                                                              # (a batch that was never searched: the gather reports this rank's failure to all)
                comms[r].gather(capi.UgsBatch(tiny, 1, 300), 0, dst=0)
            except Exception:
                pass
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join(900) for t in th]
    assert not any(t.is_alive() for t in th)
    assert not err, err
    assert all(res[r] is None for r in range(1, world))
    hits, nh, pool = res[0]
    # ---- properties over the merged table
    assert len(nh) == total_q and nh.max() <= 1 and len(hits) == int(nh.sum())
    assert len(hits) >= 0.75 * total_q
    ident_f = hits["ids"].astype(np.float64) / np.maximum(hits["aln_len"], 1)
    assert ident_f.min() >= float(np.float32(0.97)) - 1e-12
    assert np.all(hits["target"] < db.n) and np.all(np.diff(hits["query"].astype(np.int64)) > 0)      # global ids, rank order = query order
    assert np.array_equal(np.flatnonzero(nh), hits["query"])
    # paths rebased into ONE pool: shard r's offsets lie in the r-th segment of it (inside a shard the pool is filled in completion order)
    ends = hits["cigar_off"].astype(np.int64) + hits["cigar_len"]
    assert int(ends.max()) <= len(pool) and int(ends.sum() - hits["cigar_off"].astype(np.int64).sum()) == int(hits["cigar_len"].sum())
    shard_of = np.searchsorted(np.array([b[1] for b in bounds]), hits["query"], side="right")
    seg_lo = np.array([hits["cigar_off"][shard_of == r].min() for r in range(world)]); seg_hi = np.array([ends[shard_of == r].max() for r in range(world)])
    assert np.all(seg_hi[:-1] <= seg_lo[1:])
    # ---- the oracle on 300 queries of every shard (its first and last 150: both sides of every boundary)
    ids = np.array(sorted(samples))
    assert len(ids) == world * 300
    sq = np.concatenate([samples[i] for i in ids])
    so = np.zeros(len(ids) + 1, np.uint64); so[1:] = np.cumsum([len(samples[i]) for i in ids])
    odb = orc.OrcDB(orc.params(is_nucleo=True, id=0.97), db.seqs, db.offs)
    oh, onh, opool = odb.search(sq, so, nthreads=min(64, os.cpu_count() or 1))
    assert np.array_equal(nh[ids], onh)
    sel = np.flatnonzero(np.isin(hits["query"], ids))
    assert len(sel) == len(oh) > 0.7 * len(ids)
    g = hits[sel]
    assert np.array_equal(g["query"], ids[oh["query"]])
    for f in hits.dtype.names:
        if f not in ("cigar_off", "query"):
            assert np.array_equal(g[f], oh[f]), f
    for a, b in zip(g, oh):
        assert np.array_equal(pool[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["cigar_len"])],
                              opool[int(b["cigar_off"]):int(b["cigar_off"]) + int(b["cigar_len"])])
    [c.close() for c in comms]
