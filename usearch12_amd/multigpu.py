"""Multi-GPU sharding of the usearch_global path: queries are independent and the DB
is read-only, so each rank searches a contiguous query shard against its own HBM-resident replica of
the index; the ONLY exchange is one gather of the fixed-size hit tables (+counts, +path pools) to
rank 0.  Backend-agnostic: `nccl` (= RCCL over xGMI) on GPUs, `gloo` in the CPU tests."""
import numpy as np

from .abi import HIT_DTYPE


def shard_range(n_items, world, rank):
    """Contiguous shard [lo, hi) of ceil(n/world) items (the last shards may be short or empty)."""
    per = -(-n_items // world)
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def gather_tables(dist, torch, t_hits, t_n, t_pool, rank, world, dst=0):
    """Gather per-rank hit tables (uint8 views), hit counts and path pools to `dst`.
    Pools have different lengths: they are padded to the maximum over ranks (one all_reduce)."""
    dev = t_hits.device
    sizes = torch.tensor([t_hits.numel(), t_n.numel(), t_pool.numel()], dtype=torch.int64, device=dev)
    mx = sizes.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)] if rank == dst else None
    dist.gather(sizes, all_sizes, dst=dst)

    def pad(t, n):
        if t.numel() == n:
            return t.contiguous()
        out = torch.zeros(n, dtype=t.dtype, device=dev)
        out[:t.numel()] = t
        return out

    outs = []
    for t, n in ((t_hits, int(mx[0])), (t_n, int(mx[1])), (t_pool, int(mx[2]))):
        tp = pad(t, n)
        lst = [torch.empty_like(tp) for _ in range(world)] if rank == dst else None
        dist.gather(tp, lst, dst=dst)
        outs.append(lst)
    if rank != dst:
        return None
    sz = [s.cpu().numpy() for s in all_sizes]
    hits = [outs[0][r][:int(sz[r][0])].cpu().numpy() for r in range(world)]
    counts = [outs[1][r][:int(sz[r][1])].cpu().numpy() for r in range(world)]
    pools = [outs[2][r][:int(sz[r][2])].cpu().numpy() for r in range(world)]
    return hits, counts, pools


def merge_tables(hit_bytes, count_bytes, pool_bytes):
    """Rank-0 merge of the gathered per-rank results: hit_bytes[r] = compact ugs_hit records of rank r
    (uint8, global query ids already applied on the device), count_bytes[r] = uint32 hits per query,
    pool_bytes[r] = uint32 run pool.  Only the pool offsets need rebasing."""
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
