"""One rank of the world-2 CPU tests (tests/test_multigpu_gloo.py): searches its query shard with the oracle, hands its tables to the
gather over `transport` (gloo: torch.distributed; socket: bench.py's rank group) and, on rank 0, stores the merged table.
argv: transport case out_path; RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment.  Never loads libugs.so."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    transport, case, out_path = sys.argv[1:4]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    from usearch12_amd import hostgroup, multigpu
    dist = None
    if transport == "gloo":
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        group = hostgroup.GlooGroup(dist)
    else:
        group = hostgroup.SocketGroup(rank, world)
    import golden_util as G
    import orc
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
    hits["query"] += np.uint32(lo)                  # what the device grouping does with query_base
    got = multigpu.gather_tables(group, hits, nh, pool, dst=0)
    if rank == 0:
        ghits, gcnt, gpool = multigpu.merge_tables(*got)
        assert int(gcnt.sum()) == len(ghits) and len(gcnt) == qs.n
        np.save(out_path + ".hits.npy", ghits)
        np.save(out_path + ".pool.npy", gpool)
    else:
        assert got is None
    group.barrier()
    group.close()
    if dist is not None:
        dist.destroy_process_group()
    assert "usearch12_amd.capi" not in sys.modules


if __name__ == "__main__":
    main()
