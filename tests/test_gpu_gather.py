"""include/ugs_comm.h on a GPU: the gather of device-resident hit tables to one rank.
 * RCCL transport with the world a single-GPU box allows (1 rank): ncclCommInitRank from a unique id and ncclCommInitAll;
 * loopback transport with 2 and 3 ranks on one device (RCCL refuses two ranks per GPU): sizes exchange, placement in rank
   order, query ids, path-offset rebasing, HitMgr::Sort on the destination - everything except the RCCL calls themselves.
The gathered tables must equal what one ugs_batch_fetch over all queries returns."""
import threading

import numpy as np
import pytest

from usearch12_amd import capi, synth

pytestmark = pytest.mark.gpu


def _setup(both=True):
    db = synth.make_db(31, 3000, 220)
    qs = synth.make_queries(31, db, 900, 220)
    if both:
        qs = synth.revcomp_some(31, qs)
    p = capi.params(is_nucleo=True, id=0.9, max_accepts=3, max_rejects=16, strand_both=1 if both else 0)
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
    bat.upload(qs.seqs, qs.offs); bat.search(); bat.sync()
    return gdb, qs, bat.fetch()


def _runs(h, pool):
    return [tuple(pool[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["cigar_len"])]) for r in h]


def _same(got, want):
    (h, n, p), (h0, n0, p0) = got, want
    assert np.array_equal(n, n0)
    assert len(h) == len(h0)
    for f in h.dtype.names:
        if f != "cigar_off":
            assert np.array_equal(h[f], h0[f]), f
    assert _runs(h, p) == _runs(h0, p0)


def _shard_batches(gdb, qs, world):
    out = []
    for r in range(world):
        lo, hi = (qs.n * r) // world, (qs.n * (r + 1)) // world
        if r == 1:
            hi = lo                                        # an empty shard in the middle: the rank still takes part
        sub = qs.slice(lo, hi)
        b = capi.UgsBatch(gdb, max(1, sub.n), max(1, int(sub.offs[-1])))
        b.upload(sub.seqs, sub.offs); b.search(); b.sync()
        out.append((lo, hi, b))
    return out


@pytest.mark.parametrize("world,dst", [(2, 0), (3, 2)])
def test_loopback_gather_equals_one_fetch(world, dst):
    gdb, qs, want = _setup()
    shards = _shard_batches(gdb, qs, world)
    keep = np.concatenate([np.arange(lo, hi) for lo, hi, _ in shards]) if shards else np.zeros(0, int)
    comms = capi.UgsComm.init_loopback(world, 0)
    res, err = [None] * world, []

    def run(r):
        try:
            res[r] = comms[r].gather(shards[r][2], shards[r][0], dst=dst)
        except Exception as e:                               # a failing rank must not leave the others in the barrier
            err.append(e)
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join(120) for t in th]
    assert not err, err
    assert all(res[r] is None for r in range(world) if r != dst)
    h0, n0, p0 = want
    # the reference table restricted to the queries that were sharded (rank 1's shard is empty)
    mask = np.isin(h0["query"], keep)
    _same(res[dst], (h0[mask], n0[keep], p0))
    [c.close() for c in comms]


def test_rccl_world_of_one_from_unique_id_and_init_all():
    gdb, qs, want = _setup(both=False)
    bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
    bat.upload(qs.seqs, qs.offs); bat.search(); bat.sync()
    c = capi.UgsComm.init_rank(capi.UgsComm.unique_id(), 0, 1, 0)
    _same(c.gather(bat, 0), want)
    got = c.gather(bat, 0, hits_cap=len(want[0]) + 8, nq_cap=qs.n, pool_cap=len(want[2]) + 64)     # roomy buffers: one call
    _same(got, want)
    c.close()
    (c2,) = capi.UgsComm.init_all([0])
    _same(c2.gather(bat, 0), want)
    c2.close()


@pytest.mark.parametrize("name", ["hard_both", "hard_aa", "hard_acc0"])
def test_cli_gather_path_text_identical_to_reference(name, tmp_path):
    """ugs_cli's multi-GPU stage (-gpus N: one host thread per device, the rounds' hit tables gathered to rank 0 over RCCL) with the
    one device a test box has: UGS_CLI_FORCE_GATHER sends -gpus 1 through it.  Several rounds (-batch 300); byte-identical files."""
    import os
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import golden_util as G
    c, db, qs, b6, uc = G.load(name)
    cli = os.path.join(os.path.dirname(capi.LIB_PATH), "ugs_cli")
    dbfa, qfa = str(tmp_path / "db.fa"), str(tmp_path / "q.fa")
    db.write_fasta(dbfa); qs.write_fasta(qfa)
    cmd = [cli, "-usearch_global", qfa, "-db", dbfa, "-id", str(c["id"]), "-blast6out", str(tmp_path / "o.b6"), "-uc", str(tmp_path / "o.uc"),
           "-batch", "300", "-gpus", "1"]
    if not c["aa"]:
        cmd += ["-strand", c["strand"]]
    for opt in ("big", "maxaccepts", "maxrejects"):
        if opt in c:
            cmd += ["-" + opt, str(c[opt])]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL, env=dict(os.environ, UGS_CLI_FORCE_GATHER="1"))
    assert open(tmp_path / "o.b6").read() == b6
    assert open(tmp_path / "o.uc").read() == uc


def test_rccl_gather_over_every_device_of_the_box():
    """ugs_comm_init_all over ALL devices the box has: the index replicated per device, contiguous query shards,
    one host thread per rank, the grouped ncclSend / ncclRecv gather to rank 0.  On a one-GPU box this is the world of one; on an
    8-GPU node it is the first thing that runs RCCL between devices - and it must equal one fetch over all queries."""
    ndev = capi.device_count() if hasattr(capi, "device_count") else int(capi.lib().ugs_device_count())
    assert ndev >= 1
    # real multi-MB tables: ~ 110 k hit records of 80 bytes + their paths, not a golden's few hundred
    db = synth.make_db(33, 100_000, 220)
    qs = synth.make_queries(33, db, 120_000, 220)
    p = capi.params(is_nucleo=True, id=0.9, max_accepts=3, max_rejects=16)
    g0 = capi.UgsDB(p, db.seqs, db.offs, device=0)
    b0 = capi.UgsBatch(g0, qs.n, int(qs.offs[-1]))
    b0.upload(qs.seqs, qs.offs); b0.search(); b0.sync()
    want = b0.fetch()
    gdbs = [g0] + [capi.UgsDB(p, db.seqs, db.offs, device=d) for d in range(1, ndev)]
    shards = []
    for r in range(ndev):
        lo, hi = (qs.n * r) // ndev, (qs.n * (r + 1)) // ndev
        sub = qs.slice(lo, hi)
        b = capi.UgsBatch(gdbs[r], max(1, sub.n), max(1, int(sub.offs[-1])))
        b.upload(sub.seqs, sub.offs); b.search(); b.sync()
        shards.append((lo, b))
    comms = capi.UgsComm.init_all(list(range(ndev)))
    res, err = [None] * ndev, []

    def run(r):
        try:
            res[r] = comms[r].gather(shards[r][1], shards[r][0], dst=0)
        except Exception as e:
            err.append(e)
    th = [threading.Thread(target=run, args=(r,)) for r in range(ndev)]
    [t.start() for t in th]; [t.join(300) for t in th]
    assert not err, err
    assert want[0].nbytes > 6_000_000
    _same(res[0], want)
    [c.close() for c in comms]


def test_bench_ranks_over_every_device_of_the_box():
    """`bench.py --gpus ndev` with the product's gather over RCCL between PROCESSES (one rank per device) on a box with more than one GPU -
    the command the driver's scaling run issues; on a one-GPU box: the same pipeline with a communicator of one rank (--force-gather)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ndev = capi.device_count()
    args = ["--gpus", str(ndev), "--steps", "2", "--warmup", "1", "--db", "20000", "--queries", "6000", "--cpu-baseline", "none"]
    if ndev == 1:
        args.append("--force-gather")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stderr[-3000:]
    j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == ndev and "ugs_gather_results" in j["config"]["gather"]
    assert j["detail"]["hits_per_step"] > 0.6 * 6000 * ndev
    for l_ in j["detail"]["runtime_libs"]:
        assert not l_["torch_imported"] and len(l_["libamdhip64"]) == 1 and len(l_["librccl"]) == 1
