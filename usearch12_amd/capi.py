"""ctypes binding of the product library usearch12_amd/libugs.so (the C-ABI in include/ugs.h).

There is no CPU fallback: if the HIP extension is missing this module raises at load time,
and every compute entry point fails with UGS_E_NODEVICE when no gfx950 device is present.
"""
import ctypes as C
import os

import numpy as np

from .abi import FILTER_BITS, PAIR_BITS, UdbInfo, Params, HIT_DTYPE, BatchStats, ClusterStats, as_u8, XdropParams, XDROP_JOB_DTYPE, XDROP_HSP_DTYPE

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UGS_LIB") or os.path.join(HERE, "libugs.so")     # UGS_LIB: a tuning build (tools/build_variant.sh)
_lib = None

EXPORTS = [
    "ugs_params_init", "ugs_abi_version", "ugs_device_count", "ugs_device_synchronize", "ugs_debug_alloc_stats", "ugs_db_create", "ugs_db_destroy", "ugs_db_stats",
    "ugs_search_batch", "ugs_batch_create", "ugs_batch_destroy", "ugs_batch_upload", "ugs_batch_search",
    "ugs_batch_sync", "ugs_batch_wait_upload", "ugs_batch_fetch", "ugs_batch_set_query_base", "ugs_batch_get_stats", "ugs_batch_get_candidates", "ugs_batch_candidate_k", "ugs_debug_kernel_hits", "ugs_debug_rank_instances", "ugs_debug_rank_instance_name", "ugs_debug_deep_walks",
    "ugs_batch_device_results",
    "ugs_format_blast6", "ugs_format_uc_hit", "ugs_format_uc_nohit", "ugs_last_error",
    "ugs_xdrop_params_init", "ugs_xdrop_batch", "ugs_xdrop_last_stats",
    "ugs_udb_stat", "ugs_udb_read", "ugs_udb_write",
    "ugs_closedref_create", "ugs_closedref_destroy", "ugs_closedref_add", "ugs_closedref_totals", "ugs_db_set_pair_keys", "ugs_batch_set_pair_keys", "ugs_hits_sort", "ugs_params_set_local", "ugs_local_evalue", "ugs_format_blast6_local", "ugs_format_trimout", "ugs_format_userout_local", "ugs_format_alnout_header_local", "ugs_format_alnout_hit_local", "ugs_userfields_check", "ugs_format_userout", "ugs_format_blast6_nohit", "ugs_format_fasta", "ugs_hits_to_report",
    "ugs_db_masked_letters", "ugs_format_alnout_header", "ugs_format_alnout_hit", "ugs_host_register", "ugs_host_unregister",
    "ugs_format_fastapairs", "ugs_format_segout",
    "ugs_otutab_create", "ugs_otutab_destroy", "ugs_otutab_add", "ugs_otutab_write", "ugs_otutab_write_biom", "ugs_otutab_totals",
    "ugs_db_append", "ugs_params_set_cluster", "ugs_cluster_fast", "ugs_cluster_destroy", "ugs_cluster_counts", "ugs_cluster_get",
    "ugs_cluster_get_stats", "ugs_cluster_write_uc", "ugs_cluster_write_centroids",
    "ugs_cluster_fast_sorted", "ugs_label_size", "ugs_cluster_write_centroids_sized",
]


class UgsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ugs error %d: %s" % (code, msg))
        self.code = code


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("usearch12_amd/libugs.so is missing - run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
        L.ugs_params_init.argtypes = [C.POINTER(Params), i32, C.c_double]
        L.ugs_device_synchronize.argtypes = [i32]
        L.ugs_debug_alloc_stats.argtypes = [C.POINTER(C.c_ulonglong)]
        L.ugs_db_create.argtypes = [C.POINTER(Params), vp, vp, u32, i32, C.POINTER(vp)]
        L.ugs_db_destroy.argtypes = [vp]
        L.ugs_db_destroy.restype = None
        L.ugs_db_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
        L.ugs_db_debug_fetch.argtypes = [vp, vp, vp, vp]
        L.ugs_search_batch.argtypes = [vp, vp, vp, u32, vp, u64, vp, vp, u64, C.POINTER(u64)]
        L.ugs_batch_create.argtypes = [vp, u32, u64, C.POINTER(vp)]
        L.ugs_batch_destroy.argtypes = [vp]
        L.ugs_batch_destroy.restype = None
        L.ugs_batch_upload.argtypes = [vp, vp, vp, u32]
        L.ugs_batch_search.argtypes = [vp]
        L.ugs_batch_sync.argtypes = [vp]
        L.ugs_batch_fetch.argtypes = [vp, vp, u64, vp, vp, u64, C.POINTER(u64)]
        L.ugs_batch_get_stats.argtypes = [vp, C.POINTER(BatchStats)]
        L.ugs_batch_get_candidates.argtypes = [vp, vp, vp, vp, u32]
        L.ugs_batch_candidate_k.argtypes = [vp, C.POINTER(u32)]
        L.ugs_batch_set_query_base.argtypes = [vp, u32]
        L.ugs_batch_device_results.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(u64), C.POINTER(vp), C.POINTER(u64),
                                               C.POINTER(vp), C.POINTER(u64)]
        L.ugs_format_blast6.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, i32]
        L.ugs_hits_sort.argtypes = [vp, vp, u32, i32]
        L.ugs_db_set_pair_keys.argtypes = [vp, vp, vp]
        L.ugs_batch_set_pair_keys.argtypes = [vp, vp, vp]
        L.ugs_params_set_local.argtypes = [C.POINTER(Params), C.c_double, i32]
        L.ugs_local_evalue.argtypes = [C.POINTER(Params), C.c_double, u32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ugs_format_blast6_local.argtypes = [C.POINTER(Params), vp, C.c_char_p, C.c_char_p, C.c_char_p, i32]
        L.ugs_format_uc_hit.argtypes = [vp, vp, i32, C.c_char_p, C.c_char_p, C.c_char_p, i32]
        L.ugs_format_uc_nohit.argtypes = [u32, C.c_char_p, C.c_char_p, i32]
        L.ugs_last_error.restype = C.c_char_p
        L.ugs_xdrop_params_init.argtypes = [C.POINTER(XdropParams), i32]
        L.ugs_xdrop_params_init.restype = None
        L.ugs_xdrop_batch.argtypes = [i32, C.POINTER(XdropParams), vp, vp, u32, vp, vp, u32, vp, u32, vp, vp, u64, C.POINTER(u64)]
        L.ugs_xdrop_last_stats.argtypes = [C.POINTER(C.c_float), C.POINTER(u64)]
        L.ugs_host_register.argtypes = [vp, u64]
        L.ugs_host_unregister.argtypes = [vp]
        L.ugs_udb_stat.argtypes = [C.c_char_p, C.POINTER(UdbInfo)]
        L.ugs_udb_read.argtypes = [C.c_char_p, vp, vp, vp, vp, vp]
        L.ugs_udb_write.argtypes = [C.c_char_p, vp, vp, u64]
        L.ugs_db_append.argtypes = [vp, vp, vp, u32]
        L.ugs_params_set_cluster.argtypes = [C.POINTER(Params)]
        L.ugs_cluster_fast.argtypes = [C.POINTER(Params), vp, vp, u32, i32, C.POINTER(vp)]
        L.ugs_cluster_destroy.argtypes = [vp]
        L.ugs_cluster_destroy.restype = None
        L.ugs_cluster_counts.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u64), C.POINTER(u64)]
        L.ugs_cluster_get.argtypes = [vp] * 9
        L.ugs_cluster_get_stats.argtypes = [vp, C.POINTER(ClusterStats)]
        L.ugs_cluster_write_uc.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.ugs_cluster_write_centroids.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.ugs_cluster_fast_sorted.argtypes = [C.POINTER(Params), vp, vp, u32, i32, vp, i32, i32, C.POINTER(vp)]
        L.ugs_label_size.argtypes = [C.c_char_p]
        L.ugs_label_size.restype = u32
        L.ugs_cluster_write_centroids_sized.argtypes = [vp, C.c_char_p, C.c_char_p, i32, u32]
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise UgsError(rc, lib().ugs_last_error().decode())


def rank_instances():
    """(seen, compiled): bit masks of the ranking kernels this process launched so far / the library holds (include/ugs.h)"""
    a, b = C.c_uint64(0), C.c_uint64(0)
    f = lib().ugs_debug_rank_instances
    f.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]; f.restype = C.c_int
    _chk(f(C.byref(a), C.byref(b)))
    return int(a.value), int(b.value)


def rank_instance_names():
    """{bit: name} of every ranking kernel the library holds (its own table: ugs_dev.h UGS_RANK_INST_TABLE)"""
    f = lib().ugs_debug_rank_instance_name
    f.argtypes = [C.c_int]; f.restype = C.c_char_p
    return {i: f(i).decode() for i in range(64) if f(i)}


def params(is_nucleo=True, id=0.97, local_evalue=None, **kw):
    """reference defaults of usearch_global, or - with local_evalue - of usearch_local (id=None: no -id given)"""
    p = Params()
    lib().ugs_params_init(C.byref(p), 1 if is_nucleo else 0, float(0.5 if id is None else id))
    if local_evalue is not None:
        rc = lib().ugs_params_set_local(C.byref(p), float(local_evalue), 0 if id is None else 1)
        if rc != 0:
            raise UgsError(rc, last_error())
    elif id is None:                    # usearch_global without -id: ranking as for 0.5, no identity filter (accepter.cpp:35)
        p.id_set = 0
    for k, v in kw.items():
        if k in PAIR_BITS:              # pair filters of Accepter::RejectPair (-self, -minqt ...): flag or value + bit
            if v is not None and v is not False:
                p.pair_mask |= PAIR_BITS[k]
                if k not in ("self", "notself", "selfid"):
                    setattr(p, k, v)
            continue
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
        if k in FILTER_BITS:            # an optional accept filter is active when its bit is set (include/ugs.h)
            p.filter_mask |= FILTER_BITS[k]
    return p


def sort_hits(hits, counts, local=False):
    """HitMgr::Sort of a merged table (multigpu.merge_tables output) in place; returns hits"""
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    _chk(lib().ugs_hits_sort(hits.ctypes.data, counts.ctypes.data, len(counts), 1 if local else 0))
    return hits


def device_count():
    return lib().ugs_device_count()


class UgsDB:
    """Masked DB + UDB word index resident in one GPU's HBM (ugs_db)."""

    def __init__(self, p, seqs, offs, device=0):
        self.p = p
        seqs = as_u8(seqs)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        self.n = len(offs) - 1
        self.nletters = int(offs[-1])
        h = C.c_void_p()
        _chk(lib().ugs_db_create(C.byref(p), seqs.ctypes.data, offs.ctypes.data, self.n, device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().ugs_db_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def stats(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _chk(lib().ugs_db_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(postings=a.value, slots=b.value, hbm_bytes=c.value)

    def debug_fetch(self):
        st = self.stats()
        masked = np.zeros(max(self.nletters, 1), dtype=np.uint8)
        row_off = np.zeros(st["slots"] + 1, dtype=np.uint64)
        postings = np.zeros(max(st["postings"], 1), dtype=np.uint32)
        _chk(lib().ugs_db_debug_fetch(self.h, masked.ctypes.data, row_off.ctypes.data, postings.ctypes.data))
        return masked[:self.nletters], row_off, postings[:st["postings"]]

    def set_pair_keys(self, label_key=None, size=None):
        """label keys / ;size= annotations of the DB sequences for the pair filters and -abskew (ugs_db_set_pair_keys)"""
        k = None if label_key is None else np.ascontiguousarray(label_key, np.uint32)
        z = None if size is None else np.ascontiguousarray(size, np.uint32)
        _chk(lib().ugs_db_set_pair_keys(self.h, None if k is None else k.ctypes.data, None if z is None else z.ctypes.data))

    def search(self, qseqs, qoffs, pair_keys=None):
        """One-shot ugs_search_batch; with pair_keys = (query label keys, query sizes) the staged calls (the one-shot
        entry point has no room for per-query keys)."""
        qseqs = as_u8(qseqs)
        qoffs = np.ascontiguousarray(qoffs, dtype=np.uint64)
        nq = len(qoffs) - 1
        if pair_keys is not None:
            bat = UgsBatch(self, nq, int(qoffs[-1]))
            bat.upload(qseqs, qoffs)
            bat.set_pair_keys(*pair_keys)
            bat.search(); bat.sync()
            out = bat.fetch()
            bat.close()
            return out
        cap = nq * (self.p.max_accepts or 64) * (2 if self.p.strand_both else 1) * (self.p.max_hsps if self.p.local else 1) + 1
        cig_cap = int(qoffs[-1]) * 2 + 64 * nq + 1024
        for _ in range(10):
            hits = np.zeros(cap, dtype=HIT_DTYPE)
            nh = np.zeros(nq + 1, dtype=np.uint32)
            pool = np.zeros(cig_cap, dtype=np.uint32)
            used = C.c_uint64(0)
            rc = lib().ugs_search_batch(self.h, qseqs.ctypes.data, qoffs.ctypes.data, nq, hits.ctypes.data, cap,
                                        nh.ctypes.data, pool.ctypes.data, cig_cap, C.byref(used))
            if rc != -5:
                break
            # UGS_E_CAPACITY: the run pool the library asked for, or (deep walks: any number of accepts per query) a larger hit array
            if used.value > cig_cap:
                cig_cap = int(used.value) + 1024
            else:
                cap *= 4
        _chk(rc)
        nh = nh[:nq]
        return hits[:int(nh.sum())], nh, pool[:used.value]


class UgsBatch:
    """A query batch resident in HBM (ugs_batch): upload once, search many times."""

    def __init__(self, db, max_queries, max_letters):
        self.db = db
        h = C.c_void_p()
        _chk(lib().ugs_batch_create(db.h, max_queries, max_letters, C.byref(h)))
        self.h = h
        self.nq = 0
        self.nletters = 0

    def close(self):
        if getattr(self, "h", None):
            self._release_out()
            lib().ugs_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def upload(self, qseqs, qoffs):
        qseqs = as_u8(qseqs)
        qoffs = np.ascontiguousarray(qoffs, dtype=np.uint64)
        self.nq = len(qoffs) - 1
        self.nletters = int(qoffs[-1] - qoffs[0])
        self._up = qseqs            # the copy is asynchronous: the letters must stay alive until the next sync
        _chk(lib().ugs_batch_upload(self.h, qseqs.ctypes.data, qoffs.ctypes.data, self.nq))

    def sync_upload(self):
        """block until the last upload has arrived in HBM"""
        _chk(lib().ugs_batch_wait_upload(self.h))

    def set_pair_keys(self, label_key=None, size=None):
        """per-query label keys / sizes of the uploaded batch (ugs_batch_set_pair_keys)"""
        k = None if label_key is None else np.ascontiguousarray(label_key, np.uint32)
        z = None if size is None else np.ascontiguousarray(size, np.uint32)
        _chk(lib().ugs_batch_set_pair_keys(self.h, None if k is None else k.ctypes.data, None if z is None else z.ctypes.data))

    def search(self):
        _chk(lib().ugs_batch_search(self.h))

    def sync(self):
        _chk(lib().ugs_batch_sync(self.h))

    def fetch(self, reuse=False):
        """Hits of the last synced search -> (hits, nhits_per_query, run pool).  reuse=True fills result buffers owned by
        this batch (page-locked once, valid until the next fetch) instead of fresh arrays - what a streaming caller does."""
        p = self.db.p
        cap = self.nq * (p.max_accepts or 64) * (2 if p.strand_both else 1) * (p.max_hsps if p.local else 1) + 1
        cig_cap = self.nletters * 2 + 64 * self.nq + 1024
        if not reuse:
            for _ in range(10):
                hits = np.zeros(cap, dtype=HIT_DTYPE)
                nh = np.zeros(self.nq + 1, dtype=np.uint32)
                pool = np.zeros(cig_cap, dtype=np.uint32)
                used = C.c_uint64(0)
                rc = lib().ugs_batch_fetch(self.h, hits.ctypes.data, cap, nh.ctypes.data, pool.ctypes.data, cig_cap, C.byref(used))
                if rc != -5:
                    break
                # UGS_E_CAPACITY: the library reports the run pool it needs; otherwise the hit array was too small (deep walks)
                if used.value > cig_cap:
                    cig_cap = int(used.value) + 1024
                else:
                    cap *= 4
            _chk(rc)
            nh = nh[:self.nq]
            return hits[:int(nh.sum())], nh, pool[:used.value]
        bufs = getattr(self, "_out", None)
        if bufs is None or len(bufs[0]) < cap or len(bufs[1]) < self.nq + 1:
            self._release_out()
            bufs = [np.zeros(cap, dtype=HIT_DTYPE), np.zeros(self.nq + 1, dtype=np.uint32),
                    np.zeros(min(cig_cap, 24 * self.nq + 4096), dtype=np.uint32)]
            for a in bufs:
                _chk(lib().ugs_host_register(a.ctypes.data, a.nbytes))
            self._out = bufs
        while True:
            hits, nh, pool = bufs
            used = C.c_uint64(0)
            rc = lib().ugs_batch_fetch(self.h, hits.ctypes.data, len(hits), nh.ctypes.data, pool.ctypes.data, len(pool), C.byref(used))
            if rc == -5 and used.value > len(pool):               # UGS_E_CAPACITY: grow the run pool to the demanded size
                lib().ugs_host_unregister(pool.ctypes.data)
                bufs[2] = np.zeros(int(used.value * 1.25) + 4096, dtype=np.uint32)
                _chk(lib().ugs_host_register(bufs[2].ctypes.data, bufs[2].nbytes))
                continue
            _chk(rc)
            nq = self.nq
            return hits[:int(nh[:nq].sum())], nh[:nq], pool[:used.value]

    def _release_out(self):
        bufs = getattr(self, "_out", None) or []
        self._out = None
        if _lib is None:                 # interpreter shutdown: the process is about to drop the mappings anyway
            return
        for a in bufs:
            try:
                _lib.ugs_host_unregister(a.ctypes.data)
            except Exception:
                pass

    def stats(self):
        st = BatchStats()
        _chk(lib().ugs_batch_get_stats(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in BatchStats._fields_}

    def set_query_base(self, base):
        _chk(lib().ugs_batch_set_query_base(self.h, C.c_uint32(base)))

    def device_results(self, query_base=0):
        """(ptr, nbytes) of the device-resident compact hits (query ids offset by query_base),
        per-query hit counts and run pool."""
        ph, pn, pc = C.c_void_p(), C.c_void_p(), C.c_void_p()
        bh, bn, bc = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _chk(lib().ugs_batch_device_results(self.h, query_base, C.byref(ph), C.byref(bh), C.byref(pn), C.byref(bn),
                                            C.byref(pc), C.byref(bc)))
        return (ph.value, bh.value), (pn.value, bn.value), (pc.value, bc.value)

    def deep_walks(self):
        """(walks parked and continued past the 64 kept candidates, keys of their complete lists) of the last synced search"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        f = lib().ugs_debug_deep_walks
        f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]; f.restype = C.c_int
        _chk(f(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def kernel_hits(self):
        """which ranking code the last synced search ran: dict(r2_units, deferred, rank_kernel, r2_launched)"""
        out = (C.c_uint64 * 8)()
        f = lib().ugs_debug_kernel_hits
        f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]; f.restype = C.c_int
        _chk(f(self.h, out, 8))
        return {"r2_units": int(out[0]), "deferred": int(out[1]), "rank_kernel": int(out[2]), "r2_launched": int(out[3]),
                "ms_rank2": out[4] / 1000.0, "ms_rank_deferred": out[5] / 1000.0, "group_rejects": int(out[6]),
                "r2_kernel": ("", "k_rank2", "k_rank2g", "k_rank3g", "k_rank2<CL>", "k_rank2<P16>")[int(out[7])]}

    def candidates(self):
        p = self.db.p
        k = C.c_uint32(0)                # the library may keep more than max_accepts+max_rejects-1 (-selfid on the small path)
        _chk(lib().ugs_batch_candidate_k(self.h, C.byref(k)))
        K = k.value
        units = self.nq * (2 if p.strand_both else 1)
        cand = np.zeros((max(units, 1), K), dtype=np.uint32)
        cnt = np.zeros((max(units, 1), K), dtype=np.uint32)
        n = np.zeros(max(units, 1), dtype=np.uint32)
        _chk(lib().ugs_batch_get_candidates(self.h, cand.ctypes.data, cnt.ctypes.data, n.ctypes.data, K))
        return cand[:units], cnt[:units], n[:units]


def xdrop_params(is_nucleo=True, **kw):
    p = XdropParams()
    lib().ugs_xdrop_params_init(C.byref(p), 1 if is_nucleo else 0)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def xdrop_batch(p, a_set, b_set, jobs, device=0):
    """Gapped x-drop extension of every job (include/ugs.h ugs_xdrop_batch; reference XDropAlignMem /
    XDropFwdFastMem / XDropBwdFastMem).  a_set, b_set: (uint8 letters, uint64 offsets) pairs; jobs: XDROP_JOB_DTYPE
    array.  Returns (hsps[XDROP_HSP_DTYPE], path pool uint32)."""
    a_seqs, a_offs = a_set
    b_seqs, b_offs = b_set
    a_seqs, b_seqs = as_u8(a_seqs), as_u8(b_seqs)
    a_offs = np.ascontiguousarray(a_offs, dtype=np.uint64)
    b_offs = np.ascontiguousarray(b_offs, dtype=np.uint64)
    jobs = np.ascontiguousarray(jobs, dtype=XDROP_JOB_DTYPE)
    n = len(jobs)
    hsps = np.zeros(n, XDROP_HSP_DTYPE)
    cap = 1 << 16
    while True:
        pool = np.zeros(cap, np.uint32)
        used = C.c_uint64(0)
        rc = lib().ugs_xdrop_batch(device, C.byref(p), a_seqs.ctypes.data, a_offs.ctypes.data, len(a_offs) - 1,
                                   b_seqs.ctypes.data, b_offs.ctypes.data, len(b_offs) - 1, jobs.ctypes.data, n,
                                   hsps.ctypes.data, pool.ctypes.data, cap, C.byref(used))
        if rc == -5 and used.value > cap:          # UGS_E_CAPACITY: path_used holds the needed size
            cap = int(used.value)
            continue
        _chk(rc)
        return hsps, pool[:used.value]


def xdrop_last_stats():
    ms, cells = C.c_float(0), C.c_uint64(0)
    lib().ugs_xdrop_last_stats(C.byref(ms), C.byref(cells))
    return ms.value, cells.value


def udb_read(path, index=False):
    """Reference-format .udb (include/ugs.h ugs_udb_read) -> dict(is_nucleo, word_len, seqs, offs, labels[, row_sizes, postings])"""
    info = UdbInfo()
    _chk(lib().ugs_udb_stat(path.encode(), C.byref(info)))
    seqs = np.zeros(info.nletters, np.uint8)
    offs = np.zeros(info.nseq + 1, np.uint64)
    labels = np.zeros(max(info.label_bytes, 1), np.uint8)
    sizes = np.zeros(info.slots if index else 1, np.uint32)
    post = np.zeros(max(info.n_postings, 1) if index else 1, np.uint32)
    _chk(lib().ugs_udb_read(path.encode(), seqs.ctypes.data, offs.ctypes.data, labels.ctypes.data,
                            sizes.ctypes.data if index else None, post.ctypes.data if index else None))
    out = dict(is_nucleo=bool(info.is_nucleo), word_len=int(info.word_len), seqs=seqs, offs=offs,
               labels=bytes(labels[:info.label_bytes]).decode().split("\0")[:-1])
    if index:
        out.update(row_sizes=sizes, postings=post[:info.n_postings])
    return out


def udb_write(path, db, labels):
    """Write db (a UgsDB) with the given labels as a reference-format .udb (index as built on the GPU)."""
    blob = b"".join(l.encode() + b"\0" for l in labels)
    _chk(lib().ugs_udb_write(path.encode(), db.h, blob, len(blob)))


# ---- cluster_fast (include/ugs.h ugs_cluster_*)
def cluster_params(id=0.97, strand_both=False, is_nucleo=True, max_rejects=None, **kw):
    p = params(is_nucleo=is_nucleo, id=id, strand_both=1 if strand_both else 0, **kw)
    _chk(lib().ugs_params_set_cluster(C.byref(p)))
    if max_rejects is not None:              # -maxrejects on top of the command's defaults (1 .. 64)
        p.max_rejects = int(max_rejects)
    return p


class UgsCluster:
    """ugs_cluster_fast on `device`; fields as the C-ABI returns them (see include/ugs.h)"""

    SORT = {None: 0, "": 0, "length": 1, "size": 2}

    def __init__(self, p, seqs, offs, device=0, sort=None, labels=None, sizein=False):
        """sort: None | "length" | "size" (-sort); labels: the input labels (their size= annotations feed -sort size / -sizein)"""
        L = lib()
        seqs = as_u8(seqs)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        self.h = C.c_void_p()
        size_in = None
        if labels is not None:
            size_in = np.array([L.ugs_label_size(l.encode()) for l in labels], dtype=np.uint32)
        _chk(L.ugs_cluster_fast_sorted(C.byref(p), seqs.ctypes.data, offs.ctypes.data, n, self.SORT[sort],
                                       size_in.ctypes.data if size_in is not None else None, int(bool(sizein)), device, C.byref(self.h)))
        nu, nc, nh, nr = C.c_uint32(0), C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
        _chk(L.ugs_cluster_counts(self.h, C.byref(nu), C.byref(nc), C.byref(nh), C.byref(nr)))
        self.n_unique, self.n_clusters = nu.value, nc.value
        self.seq_unique = np.zeros(n, np.uint32); self.uniq_seed = np.zeros(nu.value, np.uint32)
        self.uniq_cluster = np.zeros(nu.value, np.uint32); self.uniq_nhits = np.zeros(nu.value, np.uint32)
        self.centroid_uniq = np.zeros(nc.value, np.uint32); self.cluster_size = np.zeros(nc.value, np.uint32)
        self.hits = np.zeros(nh.value, HIT_DTYPE); self.pool = np.zeros(nr.value, np.uint32)
        _chk(L.ugs_cluster_get(self.h, self.seq_unique.ctypes.data, self.uniq_seed.ctypes.data, self.uniq_cluster.ctypes.data,
                               self.uniq_nhits.ctypes.data, self.centroid_uniq.ctypes.data, self.cluster_size.ctypes.data,
                               self.hits.ctypes.data, self.pool.ctypes.data))
        self.stats = ClusterStats()
        _chk(L.ugs_cluster_get_stats(self.h, C.byref(self.stats)))

    def write_uc(self, labels, path):
        _chk(lib().ugs_cluster_write_uc(self.h, b"".join(l.encode() + b"\0" for l in labels), path.encode()))

    def write_centroids(self, labels, path, sizein=False, sizeout=False, minsize=0):
        _chk(lib().ugs_cluster_write_centroids_sized(self.h, b"".join(l.encode() + b"\0" for l in labels), path.encode(),
                                                     (1 if sizein else 0) | (2 if sizeout else 0), minsize))

    def close(self):
        if self.h:
            lib().ugs_cluster_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- include/ugs_comm.h: the RCCL gather of device-resident hit tables (libugs_rccl.so)
LIB_RCCL_PATH = os.path.join(HERE, "libugs_rccl.so")
_lib_rccl = None
COMM_EXPORTS = ["ugs_comm_unique_id", "ugs_comm_init_rank", "ugs_comm_init_all", "ugs_comm_init_loopback", "ugs_comm_destroy",
                "ugs_comm_rank", "ugs_comm_world", "ugs_gather_results", "ugs_gather_refetch", "ugs_gather_last_times"]


def lib_rccl():
    global _lib_rccl
    if _lib_rccl is None:
        lib()                                                   # libugs.so first: libugs_rccl.so resolves its symbols against it
        if not os.path.exists(LIB_RCCL_PATH):
            raise ImportError("usearch12_amd/libugs_rccl.so is missing - run __graft_entry__.build()")
        L = C.CDLL(LIB_RCCL_PATH, mode=C.RTLD_GLOBAL)
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
        L.ugs_comm_unique_id.argtypes = [C.c_char_p]
        L.ugs_comm_init_rank.argtypes = [C.c_char_p, i32, i32, i32, C.POINTER(vp)]
        L.ugs_comm_init_all.argtypes = [i32, C.POINTER(i32), C.POINTER(vp)]
        L.ugs_comm_init_loopback.argtypes = [i32, i32, C.POINTER(vp)]
        L.ugs_comm_destroy.argtypes = [vp]
        L.ugs_comm_destroy.restype = None
        L.ugs_comm_rank.argtypes = [vp]
        L.ugs_comm_world.argtypes = [vp]
        L.ugs_gather_results.argtypes = [vp, vp, u32, i32, vp, u64, vp, u64, vp, u64, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
        L.ugs_gather_refetch.argtypes = [vp, vp, u64, vp, u64, vp, u64, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
        L.ugs_gather_last_times.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib_rccl = L
    return _lib_rccl


class UgsComm:
    """one rank's communicator (include/ugs_comm.h)"""

    def __init__(self, handle):
        self.h = C.c_void_p(handle)

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _chk(lib_rccl().ugs_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def init_rank(cls, uid, rank, world, device):
        h = C.c_void_p()
        _chk(lib_rccl().ugs_comm_init_rank(uid, rank, world, device, C.byref(h)))
        return cls(h.value)

    @classmethod
    def init_all(cls, devices):
        arr = (C.c_int * len(devices))(*devices)
        hs = (C.c_void_p * len(devices))()
        _chk(lib_rccl().ugs_comm_init_all(len(devices), arr, hs))
        return [cls(h) for h in hs]

    @classmethod
    def init_loopback(cls, world, device=0):
        hs = (C.c_void_p * world)()
        _chk(lib_rccl().ugs_comm_init_loopback(world, device, hs))
        return [cls(h) for h in hs]

    def gather(self, batch, query_base, dst=0, hits_cap=0, nq_cap=0, pool_cap=0):
        """collective; on dst returns (hits, nhits_per_query, pool), elsewhere None.  Capacities 0: sized by a first call's demand."""
        L = lib_rccl()
        nh, nq, nr = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        hits = np.zeros(max(1, hits_cap), HIT_DTYPE); cnt = np.zeros(max(1, nq_cap), np.uint32); pool = np.zeros(max(1, pool_cap), np.uint32)
        rc = L.ugs_gather_results(self.h, batch.h, query_base, dst, hits.ctypes.data, hits_cap, cnt.ctypes.data, nq_cap, pool.ctypes.data, pool_cap,
                                  C.byref(nh), C.byref(nq), C.byref(nr))
        if rc == -5:                                            # UGS_E_CAPACITY on dst: the exchange is done, fetch again with room
            hits = np.zeros(max(1, nh.value), HIT_DTYPE); cnt = np.zeros(max(1, nq.value), np.uint32); pool = np.zeros(max(1, nr.value), np.uint32)
            rc = L.ugs_gather_refetch(self.h, hits.ctypes.data, len(hits), cnt.ctypes.data, len(cnt), pool.ctypes.data, len(pool),
                                      C.byref(nh), C.byref(nq), C.byref(nr))
        _chk(rc)
        if L.ugs_comm_rank(self.h) != dst:
            return None
        return hits[:nh.value], cnt[:nq.value], pool[:nr.value]

    def gather_into(self, batch, query_base, dst, hits, cnt, pool):
        """collective, into caller-owned (ideally page-locked: ugs_host_register) arrays that are reused from step to step;
        on dst returns views (hits, nhits_per_query, pool) of them, elsewhere None.  UgsError(-5) when dst's arrays are too small."""
        L = lib_rccl()
        nh, nq, nr = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        rc = L.ugs_gather_results(self.h, batch.h, query_base, dst, hits.ctypes.data, len(hits), cnt.ctypes.data, len(cnt),
                                  pool.ctypes.data, len(pool), C.byref(nh), C.byref(nq), C.byref(nr))
        self._demand = (nh.value, nq.value, nr.value)
        _chk(rc)
        if L.ugs_comm_rank(self.h) != dst:
            return None
        return hits[:nh.value], cnt[:nq.value], pool[:nr.value]

    def last_demand(self):
        """(hits, queries, runs) the last gather wanted room for on dst (what UGS_E_CAPACITY reports)"""
        return self._demand

    def refetch_into(self, hits, cnt, pool):
        """dst only, after UGS_E_CAPACITY: the tables of the last gather (still in dst's device staging buffers) once more"""
        nh, nq, nr = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _chk(lib_rccl().ugs_gather_refetch(self.h, hits.ctypes.data, len(hits), cnt.ctypes.data, len(cnt), pool.ctypes.data, len(pool),
                                           C.byref(nh), C.byref(nq), C.byref(nr)))
        return hits[:nh.value], cnt[:nq.value], pool[:nr.value]

    def last_times(self):
        a, b = C.c_double(0), C.c_double(0)
        _chk(lib_rccl().ugs_gather_last_times(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        if self.h:
            lib_rccl().ugs_comm_destroy(self.h)
            self.h = C.c_void_p()
