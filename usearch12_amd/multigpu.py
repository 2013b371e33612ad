"""Multi-GPU sharding of the usearch_global path: queries are independent and the DB
is read-only, so each rank searches a contiguous query shard against its own HBM-resident replica of
the index; the ONLY exchange is one gather of the fixed-size hit tables (+counts, +path pools) to
rank 0 - on GPUs the product's C++ gather over RCCL (include/ugs_comm.h); this module holds the shard arithmetic, the host-transport
stand-in of the gather (dry runs, CPU tests) and the rank-0 merge.  No torch, no HIP: importable anywhere."""
import numpy as np

from .abi import HIT_DTYPE


def shard_range(n_items, world, rank):
    """Contiguous shard [lo, hi) of ceil(n/world) items (the last shards may be short or empty)."""
    per = -(-n_items // world)
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def gather_tables(group, hits, counts, pool, dst=0):
    """HOST transport of the gather (a functional dry run of the N > 1 path on a box with fewer GPUs than ranks, and the CPU tests): every
    rank hands its fetched tables (numpy: ugs_hit records with GLOBAL query ids, hits per query, run pool) to `group`
    (usearch12_amd/hostgroup.py: sockets between the bench ranks, or torch.distributed gloo in the CPU tests); rank `dst` gets the
    per-rank lists (hit bytes, count bytes, pool bytes) that merge_tables takes, the others None.  The measured path is the product's own
    gather over RCCL (include/ugs_comm.h, ugs_gather.cpp) - this function never runs beside it."""
    mine = (np.ascontiguousarray(hits).view(np.uint8).reshape(-1), np.ascontiguousarray(counts, dtype=np.uint32).view(np.uint8).reshape(-1),
            np.ascontiguousarray(pool, dtype=np.uint32).view(np.uint8).reshape(-1))
    got = group.gather(mine, dst=dst)
    if got is None:
        return None
    return [g[0] for g in got], [g[1] for g in got], [g[2] for g in got]


def merge_tables(hit_bytes, count_bytes, pool_bytes, rebased=False):
    """Rank-0 merge of gathered per-rank results into one (hits[HIT_DTYPE], counts[uint32], pool[uint32]).
    Either the three concatenated arrays gather_tables returns (rebased=True: the path offsets were already rebased on
    the device, this is a re-typing of the buffers without copies) or per-rank lists (hit_bytes[r] = compact ugs_hit
    records of rank r with global query ids, count_bytes[r] = hits per query, pool_bytes[r] = run pool)."""
    if rebased:
        return (np.asarray(hit_bytes).view(np.uint8).view(HIT_DTYPE), np.asarray(count_bytes).view(np.uint8).view(np.uint32),
                np.asarray(pool_bytes).view(np.uint8).view(np.uint32))
    out_hits, out_cnt, out_pool = [], [], []
    pool_base = 0
    for r in range(len(hit_bytes)):
        h = np.frombuffer(np.ascontiguousarray(hit_bytes[r]).tobytes(), dtype=HIT_DTYPE).copy()
        pool = np.frombuffer(np.ascontiguousarray(pool_bytes[r]).tobytes(), dtype=np.uint32)
        if pool_base:
            h["cigar_off"] += np.uint64(pool_base)
        out_hits.append(h)
        out_cnt.append(np.frombuffer(np.ascontiguousarray(count_bytes[r]).tobytes(), dtype=np.uint32))
        out_pool.append(pool)
        pool_base += len(pool)
    hits = np.concatenate(out_hits) if out_hits else np.zeros(0, dtype=HIT_DTYPE)
    counts = np.concatenate(out_cnt) if out_cnt else np.zeros(0, dtype=np.uint32)
    pool = np.concatenate(out_pool) if out_pool else np.zeros(0, dtype=np.uint32)
    return hits, counts, pool
