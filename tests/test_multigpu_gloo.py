"""world_size-2 gloo test of the N>1 path on CPU: query shards, gather of fixed-size hit tables to
rank 0, merge into global hits.  The per-rank "device tables" are produced by the oracle here
(there is no GPU on this box); on GPUs bench.py feeds the same functions from ugs_batch_device_results."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_util as G
import orc
from usearch12_amd import multigpu
from usearch12_amd.abi import HIT_DTYPE, cigar_text


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, case, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if case.startswith("loc_"):                     # usearch_local: several HSPs per target, 80-byte records with raw scores
        c, db, qs, b6 = G.load_local(case)
        p = orc.params(is_nucleo=not c["aa"], **G.local_params_kw(c))
    else:
        c, db, qs, b6, uc = G.load(case)
        p = orc.params(is_nucleo=not c["aa"], id=c["id"], **G.params_kw(c))
    lo, hi = multigpu.shard_range(qs.n, world, rank)
    shard = qs.slice(lo, hi)
    odb = orc.OrcDB(p, db.seqs, db.offs)            # replica of the index on every rank
    hits, nh, pool = odb.search(shard.seqs, shard.offs)
    hits = hits.copy()
    hits["query"] += np.uint32(lo)                  # what the device compaction does with query_base
    t_h = torch.from_numpy(hits.view(np.uint8).reshape(-1).copy())
    t_n = torch.from_numpy(nh.astype(np.uint32).view(np.uint8).copy())
    t_p = torch.from_numpy(pool.astype(np.uint32).view(np.uint8).copy())
    got = multigpu.gather_tables(dist, torch, t_h, t_n, t_p, rank, world, dst=0)
    if rank == 0:
        ghits, gcnt, gpool = multigpu.merge_tables(got[0], got[1], got[2], rebased=got[3])
        assert int(gcnt.sum()) == len(ghits) and len(gcnt) == qs.n
        np.save(out_path + ".hits.npy", ghits)
        np.save(out_path + ".pool.npy", gpool)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_run(tmp_path):
    case = "hard_acc"          # maxaccepts 4: several hits per query
    out = str(tmp_path / "g")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, case, out), nprocs=2, join=True)
    ghits = np.load(out + ".hits.npy")
    gpool = np.load(out + ".pool.npy")
    c, db, qs, b6, uc = G.load(case)
    p = orc.params(is_nucleo=True, id=c["id"], **G.params_kw(c))
    hits, nh, pool = orc.OrcDB(p, db.seqs, db.offs).search(qs.seqs, qs.offs)
    # per-query order inside a shard is the oracle's; compare as (query, target, path) multisets in order
    assert len(ghits) == len(hits)
    q_single = np.repeat(np.arange(qs.n), nh)
    assert np.array_equal(ghits["query"], q_single)
    for f in ("target", "ids", "mism", "aln_len", "opens", "qlo", "qhi", "tlo", "thi", "strand", "cigar_len"):
        assert np.array_equal(ghits[f], hits[f]), f
    for a, b in zip(ghits[::37], hits[::37]):
        assert cigar_text(gpool, a["cigar_off"], a["cigar_len"]) == cigar_text(pool, b["cigar_off"], b["cigar_len"])


def test_two_rank_gather_local_hits(tmp_path):
    case = "loc_nt_both"
    out = str(tmp_path / "g")
    mp.spawn(_worker, args=(2, _free_port(), case, out), nprocs=2, join=True)
    ghits = np.load(out + ".hits.npy")
    gpool = np.load(out + ".pool.npy")
    c, db, qs, b6 = G.load_local(case)
    p = orc.params(is_nucleo=True, **G.local_params_kw(c))
    hits, nh, pool = orc.OrcDB(p, db.seqs, db.offs).search(qs.seqs, qs.offs)
    assert len(ghits) == len(hits) and np.all(ghits["flags"] & 1 == 1)
    for f in hits.dtype.names:
        if f != "cigar_off":
            assert np.array_equal(ghits[f], hits[f]), f
    got = orc.format_blast6_local(orc.lib(), "orc", p, ghits, nh, qs.labels(), db.labels())
    assert got == b6
    for a, b in zip(ghits[::29], hits[::29]):
        assert cigar_text(gpool, a["cigar_off"], a["cigar_len"]) == cigar_text(pool, b["cigar_off"], b["cigar_len"])


def test_hits_sort_orders_a_candidate_order_table():
    """ugs_hits_sort (host-only ABI entry): per query non-increasing score, same records; on a table that is already
    in HitMgr order and has no tied scores it is the identity"""
    from usearch12_amd import capi
    c, db, qs, b6 = G.load_local("loc_aa_acc")
    p = orc.params(is_nucleo=False, **G.local_params_kw(c))
    hits, nh, pool = orc.OrcDB(p, db.seqs, db.offs).search(qs.seqs, qs.offs)
    rng = np.random.default_rng(3)
    shuffled = hits.copy()
    k = 0
    for n in nh:
        n = int(n)
        shuffled[k:k + n] = shuffled[k:k + n][rng.permutation(n)]
        k += n
    out = capi.sort_hits(shuffled.copy(), nh, local=True)
    k = 0
    for n in nh:
        n = int(n)
        a, b = out[k:k + n], hits[k:k + n]
        assert np.all(np.diff(a["raw_score"]) <= 0)
        assert sorted(a.tobytes()[i * 80:(i + 1) * 80] for i in range(n)) == sorted(b.tobytes()[i * 80:(i + 1) * 80] for i in range(n))
        if len(set(b["raw_score"].tolist())) == n:
            assert a.tobytes() == b.tobytes()
        k += n


def test_shard_ranges_cover():
    for n in (0, 1, 7, 10, 1000003):
        for w in (1, 2, 3, 8):
            r = [multigpu.shard_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
