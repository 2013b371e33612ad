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
    Pools have different lengths: they are padded to the maximum over ranks (one all_reduce).
    On `dst` returns (hit bytes, count bytes, pool bytes, True): the three tables concatenated in rank order ON THE
    DEVICE, every rank's path offsets rebased there, and each copied to the host once into a reusable (page-locked,
    for GPU tensors) buffer that stays valid until the next call.  Other ranks get None."""
    dev = t_hits.device
    sizes = torch.tensor([t_hits.numel(), t_n.numel(), t_pool.numel()], dtype=torch.int64, device=dev)
    mx = sizes.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)] if rank == dst else None
    dist.gather(sizes, all_sizes, dst=dst)

    def pad(t, n):
        # always a torch-owned staging copy: the inputs alias the library's device buffers, which the next search rewrites
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
        if t_hits.is_cuda:
            torch.cuda.synchronize()                                # the sends have left the staging copies
        return None
    sz = torch.stack(all_sizes).cpu().numpy()                       # [world, 3]
    res = []
    for k in range(3):
        parts = [outs[k][r][:int(sz[r][k])] for r in range(world)]
        cat = torch.cat(parts) if world > 1 else parts[0].clone()    # concatenated on the device: ONE copy to the host per table
        if k == 0 and cat.dtype == torch.uint8 and cat.numel() % HIT_DTYPE.itemsize == 0 and cat.numel():
            # rebase every rank's path offsets (ugs_hit.cigar_off, byte 56 = int64 column 7) to the concatenated pool,
            # on the device, so that the merged table needs no per-record work on the host
            rec = cat.view(torch.int64).view(-1, HIT_DTYPE.itemsize // 8)
            lo, base = 0, 0
            for r in range(world):
                n = int(sz[r][0]) // HIT_DTYPE.itemsize
                if base:
                    rec[lo:lo + n, 7] += base
                lo += n
                base += int(sz[r][2]) // 4 if t_pool.dtype == torch.uint8 else int(sz[r][2])   # pool entries are uint32 runs
        host = _host_buffer(torch, k, cat)
        host[:cat.numel()].copy_(cat, non_blocking=cat.is_cuda)
        res.append((host, cat.numel()))
    if t_hits.is_cuda:
        torch.cuda.synchronize()
    return tuple(h.numpy()[:n] for h, n in res) + (True,)              # (hits bytes, counts, pool, already_rebased)


_HOST = {}


def _host_buffer(torch, k, t):
    """Reusable result buffer on the host, page-locked when the source lives on a GPU (valid until the next gather)."""
    buf = _HOST.get((k, t.dtype))
    if buf is None or buf.numel() < t.numel():
        n = int(t.numel() * 1.25) + 1024
        buf = torch.empty(n, dtype=t.dtype, pin_memory=True) if t.is_cuda else torch.empty(n, dtype=t.dtype)
        _HOST[(k, t.dtype)] = buf
    return buf


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
