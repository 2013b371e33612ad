"""world_size-2 tests of the N > 1 path on CPU: query shards, gather of the per-rank hit tables to rank 0, merge into global hits.
The per-rank "device tables" are produced by the oracle here (there is no GPU on this box).  Two transports carry them:
  * torch.distributed with the gloo backend (usearch12_amd/hostgroup.py GlooGroup) - in CHILD processes (tests/gloo_worker.py) that
    never load libugs.so: torch's bundled HIP runtime and the system one must not meet in one process;
  * the socket group bench.py's ranks use (SocketGroup), in multiprocessing children.
On GPUs the tables travel through the product's C++ gather (ugs_gather.cpp; tests/test_gpu_gather.py)."""
import os
import socket
import subprocess
import sys

import numpy as np

import golden_util as G
import orc
from usearch12_amd import multigpu
from usearch12_amd.abi import cigar_text

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _spawn(world, case, out, transport="gloo"):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "gloo_worker.py"), transport, case, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        log = p.communicate(timeout=600)[0]
        assert p.returncode == 0, log[-3000:]


def test_two_rank_gather_equals_single_run(tmp_path):
    case = "hard_acc"          # maxaccepts 4: several hits per query
    out = str(tmp_path / "g")
    _spawn(2, case, out)
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
    _spawn(2, case, out)
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


def test_socket_group_three_ranks_equal_gloo(tmp_path):
    """bench.py's own rank group (sockets, no torch) carries the same tables to the same merged result, with an odd world"""
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    _spawn(3, "hard_acc", a, transport="socket")
    _spawn(3, "hard_acc", b, transport="gloo")
    for suffix in (".hits.npy", ".pool.npy"):
        assert np.array_equal(np.load(a + suffix), np.load(b + suffix))


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
