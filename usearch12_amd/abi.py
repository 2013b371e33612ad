"""ctypes mirror of include/ugs.h (structs + error codes), shared by the product
binding (capi.py) and the test-only oracle binding (tests/orc.py)."""
import ctypes as C
import numpy as np

UGS_OK = 0
UGS_E_ARG, UGS_E_NODEVICE, UGS_E_HIP, UGS_E_NOMEM, UGS_E_CAPACITY, UGS_E_ENVELOPE = -1, -2, -3, -4, -5, -6


class Params(C.Structure):
    _fields_ = [
        ("is_nucleo", C.c_int32), ("word_len", C.c_int32), ("id", C.c_float),
        ("id_accept", C.c_double), ("id_set", C.c_int32), ("strand_both", C.c_int32),
        ("max_accepts", C.c_int32), ("max_rejects", C.c_int32),
        ("big", C.c_uint32), ("bump_pct", C.c_uint32), ("stepwords", C.c_uint32),
        ("band", C.c_int32), ("minhsp", C.c_int32), ("xdrop_nw", C.c_float),
        ("match", C.c_float), ("mismatch", C.c_float),
        ("hsp_word_len", C.c_int32), ("dbmask", C.c_int32),
        ("filter_mask", C.c_uint32),
        ("maxid", C.c_float), ("query_cov", C.c_float), ("max_query_cov", C.c_float), ("target_cov", C.c_float),
        ("max_target_cov", C.c_float),
        ("mincols", C.c_uint32), ("maxgaps", C.c_uint32), ("maxdiffs", C.c_uint32), ("mindiffs", C.c_uint32),
        ("local", C.c_int32), ("evalue", C.c_float), ("xdrop_u", C.c_float), ("xdrop_g", C.c_float),
        ("local_open", C.c_float), ("local_ext", C.c_float), ("ka_dbsize", C.c_float), ("max_hsps", C.c_uint32),
        ("pair_mask", C.c_uint32), ("min_sizeratio", C.c_float), ("minqt", C.c_float), ("maxqt", C.c_float), ("minsl", C.c_float),
        ("maxsl", C.c_float), ("abskew", C.c_float),
        ("align_flags", C.c_uint32), ("termid", C.c_float), ("termidd", C.c_float),
    ]


# UGS_F_* bits of Params.filter_mask (include/ugs.h)
F_MAXID, F_MINCOLS, F_MAXGAPS, F_QUERY_COV, F_MAX_QUERY_COV, F_TARGET_COV, F_MAX_TARGET_COV, F_MAXDIFFS, F_MINDIFFS, F_ABSKEW = (
    1, 2, 4, 8, 16, 32, 64, 128, 256, 512)
FILTER_BITS = dict(maxid=F_MAXID, mincols=F_MINCOLS, maxgaps=F_MAXGAPS, query_cov=F_QUERY_COV, max_query_cov=F_MAX_QUERY_COV,
                   target_cov=F_TARGET_COV, max_target_cov=F_MAX_TARGET_COV, maxdiffs=F_MAXDIFFS, mindiffs=F_MINDIFFS, abskew=F_ABSKEW)
# UGS_P_* bits of Params.pair_mask; self/notself/selfid are flags (pass True), the others carry a value
A_FULLDP, A_GAFORCE, A_TERMID, A_TERMIDD = 1, 2, 4, 8
PAIR_BITS = dict(self=1, notself=2, selfid=4, min_sizeratio=8, minqt=16, maxqt=32, minsl=64, maxsl=128)


HIT_DTYPE = np.dtype({
    "names": ["query", "target", "ids", "mism", "gaps_int", "aln_len", "opens",
              "qlo", "qhi", "tlo", "thi", "ql", "tl", "strand", "cigar_off", "cigar_len", "cols", "raw_score", "flags"],
    "formats": ["<u4"] * 14 + ["<u8", "<u4", "<u4", "<f4", "<u4"],
    "offsets": [0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48, 52, 56, 64, 68, 72, 76],
    "itemsize": 80,
})
HIT_LOCAL = 1
HIT_ORDER_SHIFT = 8        # flags >> 8: position in HitMgr's append order within the query


class BatchStats(C.Structure):
    _fields_ = [
        ("ms_rank", C.c_float), ("ms_align", C.c_float), ("ms_total", C.c_float),
        ("postings", C.c_uint64), ("query_letters", C.c_uint64), ("target_letters", C.c_uint64),
        ("pairs_aligned", C.c_uint64), ("dp_cells", C.c_uint64), ("hits", C.c_uint64),
        ("ms_rank_setup", C.c_float), ("reserved_", C.c_float),
    ]


class ClusterStats(C.Structure):
    """mirror of ugs_cluster_stats (include/ugs.h)"""
    _fields_ = [("batches", C.c_uint32), ("batches_cut", C.c_uint32), ("max_batch", C.c_uint32), ("units_heavy", C.c_uint32),
                ("queries_redone", C.c_uint64), ("inbatch_entries", C.c_uint64), ("pairs_in_batch", C.c_uint64),
                ("hits_in_batch", C.c_uint64), ("pairs_frozen", C.c_uint64), ("postings", C.c_uint64),
                ("ms_rank", C.c_float), ("ms_align", C.c_float)] + \
               [(n, C.c_float) for n in ("s_derep", "s_search", "s_inbatch", "s_d2h", "s_replay", "s_pairs", "s_append", "s_total")]


class UdbInfo(C.Structure):
    """mirror of ugs_udb_info (include/ugs.h)"""
    _fields_ = [("is_nucleo", C.c_int32), ("word_len", C.c_uint32), ("nseq", C.c_uint64), ("nletters", C.c_uint64),
                ("label_bytes", C.c_uint64), ("slots", C.c_uint64), ("n_postings", C.c_uint64)]


class XdropParams(C.Structure):
    """mirror of ugs_xdrop_params (include/ugs.h)"""
    _fields_ = [("is_nucleo", C.c_int32), ("match", C.c_float), ("mismatch", C.c_float),
                ("local_open", C.c_float), ("local_ext", C.c_float), ("xdrop", C.c_float)]


XDROP_ALIGN, XDROP_FWD, XDROP_BWD = 0, 1, 2
XDROP_JOB_DTYPE = np.dtype([("a", "<u4"), ("b", "<u4"), ("anc_loi", "<u4"), ("anc_loj", "<u4"), ("anc_len", "<u4"),
                            ("mode", "<u4")])
XDROP_HSP_DTYPE = np.dtype([("score", "<f4"), ("loi", "<u4"), ("loj", "<u4"), ("leni", "<u4"), ("lenj", "<u4"),
                            ("path_len", "<u4"), ("path_off", "<u8")])
assert XDROP_JOB_DTYPE.itemsize == 24 and XDROP_HSP_DTYPE.itemsize == 32


def path_text(pool, off, n):
    """run-length pool (len<<2|op) -> M/D/I text"""
    runs = pool[int(off):int(off) + int(n)]
    return "".join("MDI"[int(r) & 3] * (int(r) >> 2) for r in runs)


def as_u8(buf):
    a = np.frombuffer(buf, dtype=np.uint8) if isinstance(buf, (bytes, bytearray)) else np.ascontiguousarray(buf, dtype=np.uint8)
    return a


def ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def cigar_text(pool, off, n):
    """CompressPath text (comppath.cpp:7-48) from run-length pool entries."""
    out = []
    for r in pool[int(off):int(off) + int(n)]:
        ln, op = int(r) >> 2, "MDI"[int(r) & 3]
        out.append(op if ln == 1 else "%d%s" % (ln, op))
    return "".join(out)


def read_fasta(path):
    """Minimal FASTA reader with the reference's filtering rules
    (fastaseqsource.cpp:25-124): labels keep everything after '>', whitespace and
    '-'/'.' are stripped from sequences, other non-alpha bytes dropped, empty records skipped."""
    labels, seqs = [], []
    cur = None
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if cur is not None and cur[1]:
                    labels.append(cur[0]); seqs.append(b"".join(cur[1]))
                cur = (line[1:].decode(), [])
            elif cur is not None:
                s = bytes(c for c in line if (65 <= c <= 90) or (97 <= c <= 122))
                if s:
                    cur[1].append(s)
    if cur is not None and cur[1]:
        labels.append(cur[0]); seqs.append(b"".join(cur[1]))
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    flat = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy() if seqs else np.zeros(0, np.uint8)
    return labels, flat, offs
