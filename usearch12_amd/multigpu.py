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


def merge_tables(hit_bytes, count_bytes, pool_bytes, shard_lo, max_accepts):
    """Rank-0 merge of gathered device tables into one compact hit array with GLOBAL query indexes and
    pool offsets.  hit_bytes[r] is the raw [units*max_accepts] ugs_hit table of rank r (uint8),
    count_bytes[r] its uint32 per-unit hit counts, pool_bytes[r] its uint32 run pool (as uint8)."""
    out_hits, out_pool = [], []
    pool_base = 0
    for r in range(len(hit_bytes)):
        cnt = np.frombuffer(np.ascontiguousarray(count_bytes[r]).tobytes(), dtype=np.uint32)
        tab = np.frombuffer(np.ascontiguousarray(hit_bytes[r]).tobytes(), dtype=HIT_DTYPE)
        pool = np.frombuffer(np.ascontiguousarray(pool_bytes[r]).tobytes(), dtype=np.uint32)
        units = len(cnt)
        if units:
            tab = tab[:units * max_accepts].reshape(units, max_accepts)
            mask = np.arange(max_accepts)[None, :] < cnt[:, None]
            h = tab[mask].copy()
            h["query"] += np.uint32(shard_lo[r])
            h["cigar_off"] += np.uint64(pool_base)
            out_hits.append(h)
        out_pool.append(pool)
        pool_base += len(pool)
    hits = np.concatenate(out_hits) if out_hits else np.zeros(0, dtype=HIT_DTYPE)
    pool = np.concatenate(out_pool) if out_pool else np.zeros(0, dtype=np.uint32)
    return hits, pool
