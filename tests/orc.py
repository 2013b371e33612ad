"""TEST-ONLY ctypes binding of oracle/liborc.so (the CPU restatement used as parity checker)."""
import ctypes as C
import os
import subprocess
import numpy as np

from usearch12_amd.abi import FILTER_BITS, PAIR_BITS, Params, HIT_DTYPE, ptr, as_u8, cigar_text, XdropParams, XDROP_JOB_DTYPE, XDROP_HSP_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
REF_BIN = os.path.join(ORC_DIR, "_ref", "usearch12")
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORC_DIR, "liborc.so")
        src = os.path.join(ORC_DIR, "ugs_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORC_DIR, "liborc.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(so)
        L.orc_db_create.restype = C.c_int
        L.orc_db_create.argtypes = [C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.orc_db_destroy.argtypes = [C.c_void_p]
        L.orc_db_masked.restype = C.c_void_p
        L.orc_db_masked.argtypes = [C.c_void_p]
        L.orc_db_slots.restype = C.c_uint64
        L.orc_db_slots.argtypes = [C.c_void_p]
        L.orc_db_row_off.restype = C.c_void_p
        L.orc_db_row_off.argtypes = [C.c_void_p]
        L.orc_db_postings.restype = C.c_void_p
        L.orc_db_postings.argtypes = [C.c_void_p]
        L.orc_search_batch.restype = C.c_int
        L.orc_search_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64,
                                       C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_int]
        L.orc_rank.restype = C.c_int
        L.orc_rank.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_align_pair.restype = C.c_int
        L.orc_align_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                     C.c_uint32, C.POINTER(C.c_float)]
        L.orc_viterbi_band.restype = C.c_float
        L.orc_viterbi_band.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                       C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]
        L.orc_fastmask.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_revcomp.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_params_init.argtypes = [C.POINTER(Params), C.c_int, C.c_double]
        L.orc_get_stats.argtypes = [C.c_void_p, C.c_void_p]
        for fn in (L.orc_format_blast6,):
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.orc_format_uc_hit.restype = C.c_int
        L.orc_format_uc_hit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.orc_format_uc_nohit.restype = C.c_int
        L.orc_format_uc_nohit.argtypes = [C.c_uint32, C.c_char_p, C.c_char_p, C.c_int]
        L.orc_db_set_pair_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_set_query_pair_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_params_set_local.argtypes = [C.POINTER(Params), C.c_double, C.c_int]
        L.orc_local_evalue.argtypes = [C.POINTER(Params), C.c_double, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_local_rescore_diffs.restype = C.c_ulong
        L.orc_format_blast6_local.restype = C.c_int
        L.orc_format_blast6_local.argtypes = [C.POINTER(Params), C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.orc_xdrop_params_init.argtypes = [C.POINTER(XdropParams), C.c_int]
        L.orc_xdrop_job.restype = C.c_int
        L.orc_xdrop_job.argtypes = [C.POINTER(XdropParams), C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_void_p,
                                    C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def xdrop_params(is_nucleo=True, xdrop=32.0):
    p = XdropParams()
    lib().orc_xdrop_params_init(C.byref(p), int(is_nucleo))
    p.xdrop = xdrop
    return p


def xdrop_job(p, a, b, mode, anc=(0, 0, 0)):
    """one x-drop job through the oracle -> (score, loi, loj, leni, lenj, path text, cells) or None if rc<0"""
    a = a if isinstance(a, bytes) else a.encode()
    b = b if isinstance(b, bytes) else b.encode()
    job = np.zeros(1, XDROP_JOB_DTYPE)
    job["anc_loi"], job["anc_loj"], job["anc_len"] = anc
    job["mode"] = mode
    hsp = np.zeros(1, XDROP_HSP_DTYPE)
    path = C.create_string_buffer(len(a) + len(b) + 8)
    cells = C.c_uint64(0)
    rc = lib().orc_xdrop_job(C.byref(p), a, len(a), b, len(b), job.ctypes.data, hsp.ctypes.data, path, C.byref(cells))
    if rc < 0:
        return None
    h = hsp[0]
    return float(h["score"]), int(h["loi"]), int(h["loj"]), int(h["leni"]), int(h["lenj"]), path.value.decode(), cells.value


def params(is_nucleo=True, id=0.97, local_evalue=None, **kw):
    """id=None with local_evalue set: usearch_local without -id"""
    p = Params()
    lib().orc_params_init(C.byref(p), 1 if is_nucleo else 0, float(0.5 if id is None else id))
    if local_evalue is not None:
        lib().orc_params_set_local(C.byref(p), float(local_evalue), 0 if id is None else 1)
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


STATS_DTYPE = np.dtype([(n, "<u8") for n in ("postings", "query_letters", "target_letters", "pairs_aligned",
                                             "dp_cells", "hits", "ungapped_calls", "dp_calls")])


class OrcDB:
    def __init__(self, p, seqs, offs):
        self.p = p
        self.seqs = as_u8(seqs)
        self.offs = np.ascontiguousarray(offs, dtype=np.uint64)
        self.n = len(self.offs) - 1
        h = C.c_void_p()
        rc = lib().orc_db_create(C.byref(p), self.seqs.ctypes.data, self.offs.ctypes.data, self.n, C.byref(h))
        assert rc == 0, rc
        self.h = h

    def close(self):
        if self.h:
            lib().orc_db_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def masked(self):
        n = int(self.offs[-1])
        return np.ctypeslib.as_array(C.cast(lib().orc_db_masked(self.h), C.POINTER(C.c_uint8)), shape=(n,)).copy()

    def index(self):
        slots = int(lib().orc_db_slots(self.h))
        ro = np.ctypeslib.as_array(C.cast(lib().orc_db_row_off(self.h), C.POINTER(C.c_uint64)), shape=(slots + 1,)).copy()
        n = int(ro[-1])
        po = np.ctypeslib.as_array(C.cast(lib().orc_db_postings(self.h), C.POINTER(C.c_uint32)), shape=(max(n, 1),))[:n].copy()
        return ro, po

    def set_pair_keys(self, label_key, size):
        self._tk = np.ascontiguousarray(label_key, np.uint32); self._tz = np.ascontiguousarray(size, np.uint32)
        lib().orc_db_set_pair_keys(self.h, self._tk.ctypes.data, self._tz.ctypes.data)

    def set_query_pair_keys(self, label_key, size):
        self._qk = np.ascontiguousarray(label_key, np.uint32); self._qz = np.ascontiguousarray(size, np.uint32)
        lib().orc_set_query_pair_keys(self.h, self._qk.ctypes.data, self._qz.ctypes.data)

    def search(self, qseqs, qoffs, nthreads=1):
        qseqs = as_u8(qseqs)
        qoffs = np.ascontiguousarray(qoffs, dtype=np.uint64)
        nq = len(qoffs) - 1
        cap = nq * (self.p.max_accepts or 64) * (2 if self.p.strand_both else 1) * (64 if self.p.local else 1) + 1
        cig_cap = (int(qoffs[-1]) * 2 + 64 * nq + 1024) * (8 if self.p.local else 1)
        for _ in range(8):                                  # (unlimited accepts: the caller's buffers grow until the hits fit)
            hits = np.zeros(cap, dtype=HIT_DTYPE)
            nh = np.zeros(nq + 1, dtype=np.uint32)
            pool = np.zeros(cig_cap, dtype=np.uint32)
            used = C.c_uint64(0)
            rc = lib().orc_search_batch(self.h, qseqs.ctypes.data, qoffs.ctypes.data, nq, hits.ctypes.data, cap,
                                        nh.ctypes.data, pool.ctypes.data, cig_cap, C.byref(used), nthreads)
            if rc != -5:
                break
            cap *= 4; cig_cap *= 4
        assert rc == 0, rc
        nh = nh[:nq]
        return hits[:int(nh.sum())], nh, pool[:used.value]

    def stats(self):
        st = np.zeros(1, dtype=STATS_DTYPE)
        lib().orc_get_stats(self.h, st.ctypes.data)
        return {k: int(st[0][k]) for k in STATS_DTYPE.names}

    def rank(self, q, cap=64):
        q = as_u8(q)
        cand = np.zeros(cap, dtype=np.uint32)
        cnt = np.zeros(cap, dtype=np.uint32)
        n = lib().orc_rank(self.h, q.ctypes.data, len(q), cand.ctypes.data, cnt.ctypes.data, cap)
        m = min(n, cap)
        return n, cand[:m], cnt[:m]

    def align_pair(self, q, t):
        q, t = as_u8(q), as_u8(t)
        buf = C.create_string_buffer(len(q) + len(t) + 8)
        fid = C.c_float(0)
        ok = lib().orc_align_pair(self.h, q.ctypes.data, len(q), t.ctypes.data, len(t), buf, len(buf), C.byref(fid))
        return ok, buf.value.decode(), fid.value

    def viterbi(self, a, b, band, pen):
        a, b = as_u8(a), as_u8(b)
        pen = np.ascontiguousarray(pen, dtype=np.float32)
        buf = C.create_string_buffer(len(a) + len(b) + 8)
        cells = C.c_uint64(0)
        s = lib().orc_viterbi_band(self.h, a.ctypes.data, len(a), b.ctypes.data, len(b), band, pen.ctypes.data,
                                   buf, len(buf), C.byref(cells))
        return s, buf.value.decode(), cells.value


def revcomp(seq):
    s = as_u8(seq)
    out = np.zeros(len(s), dtype=np.uint8)
    lib().orc_revcomp(s.ctypes.data, len(s), out.ctypes.data)
    return out


def fastmask(seq):
    s = as_u8(seq).copy()
    lib().orc_fastmask(s.ctypes.data, len(s))
    return s


def format_outputs(fmt_lib, prefix, hits, nh, pool, qlabels, qlens, tlabels, is_nucleo):
    """Render blast6 and uc text in query order with the writers of `fmt_lib`
    (prefix 'orc' for the oracle, 'ugs' for the product)."""
    b6, uc = [], []
    buf = C.create_string_buffer(1 << 16)
    f_b6 = getattr(fmt_lib, prefix + "_format_blast6")
    f_uch = getattr(fmt_lib, prefix + "_format_uc_hit")
    f_ucn = getattr(fmt_lib, prefix + "_format_uc_nohit")
    pool = np.ascontiguousarray(pool, dtype=np.uint32)
    k = 0
    for qi, n in enumerate(nh):
        ql = qlabels[qi].encode()
        if n == 0:
            f_ucn(int(qlens[qi]), ql, buf, len(buf))
            uc.append(buf.value.decode())
        for j in range(int(n)):
            h = hits[k:k + 1]
            tl = tlabels[int(h["target"][0])].encode()
            f_b6(h.ctypes.data, ql, tl, buf, len(buf))
            b6.append(buf.value.decode())
            f_uch(h.ctypes.data, pool.ctypes.data, 1 if is_nucleo else 0, ql, tl, buf, len(buf))
            uc.append(buf.value.decode())
            k += 1
    return "".join(b6), "".join(uc)


def run_reference(qfa, dbfa, out_prefix, id=0.97, strand="plus", threads=1, extra=()):
    """Run the compiled unmodified reference (oracle/_ref/usearch12) - only where it exists."""
    cmd = [REF_BIN, "-usearch_global", qfa, "-db", dbfa, "-id", str(id), "-blast6out", out_prefix + ".b6",
           "-uc", out_prefix + ".uc", "-threads", str(threads)]
    if strand:
        cmd += ["-strand", strand]
    cmd += list(extra)
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out_prefix + ".b6").read(), open(out_prefix + ".uc").read()


def format_blast6_local(fmt_lib, prefix, p, hits, nh, qlabels, tlabels):
    """blast6 text of usearch_local hits in query order ('orc' = oracle writer, 'ugs' = product writer)"""
    out = []
    buf = C.create_string_buffer(1 << 16)
    f = getattr(fmt_lib, prefix + "_format_blast6_local")
    k = 0
    for qi, n in enumerate(nh):
        ql = qlabels[qi].encode()
        for j in range(int(n)):
            h = hits[k:k + 1]
            f(C.byref(p), h.ctypes.data, ql, tlabels[int(h["target"][0])].encode(), buf, len(buf))
            out.append(buf.value.decode())
            k += 1
    return "".join(out)


# ---- cluster_fast (oracle/ugs_oracle.c orc_cluster_fast)
def _vp(a):
    return a.ctypes.data


def cluster_params(id=0.97, strand_both=False, is_nucleo=True, max_rejects=None, **kw):
    """cmd_cluster_fast's searcher: terminator 1 accept / 8 rejects (terminator.cpp:10-14), letters used as read"""
    return params(is_nucleo=is_nucleo, id=id, max_accepts=1, max_rejects=8 if max_rejects is None else int(max_rejects), dbmask=2,
                  strand_both=1 if strand_both else 0, **kw)


class ClusterResult:
    pass


SORT_MODES = {None: 0, "": 0, "length": 1, "size": 2}


def label_sizes(labels):
    """GetSizeFromLabel (label.cpp:152-161) per label: (unsigned) atoi after the first ";size=", 0xffffffff = no annotation"""
    import re
    out = np.full(len(labels), 0xFFFFFFFF, np.uint32)
    for i, l in enumerate(labels):
        k = l.find(";size=")
        if k >= 0:
            m = re.match(r"\s*[+-]?\d+", l[k + 6:])
            out[i] = (int(m.group(0)) if m else 0) & 0xFFFFFFFF
    return out


def cluster_fast(p, seqs, offs, sort=None, size_in=None, sizein=False):
    L = lib()
    seqs = as_u8(seqs)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    n = len(offs) - 1
    if size_in is not None:
        size_in = np.ascontiguousarray(size_in, dtype=np.uint32)
    L.orc_cluster_fast_sorted.restype = C.c_int
    L.orc_cluster_fast_sorted.argtypes = [C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 2 + \
        [C.POINTER(C.c_uint32)] + [C.c_void_p] * 4 + [C.POINTER(C.c_uint32), C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                                      C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    r = ClusterResult()
    r.seq_unique = np.zeros(n, np.uint32); r.uniq_seed = np.zeros(n, np.uint32)
    r.uniq_cluster = np.zeros(n, np.uint32); r.uniq_nhits = np.zeros(n, np.uint32)
    r.centroid_uniq = np.zeros(n, np.uint32); r.cluster_size = np.zeros(n, np.uint32)
    nu, nc, nh, cu = C.c_uint32(0), C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
    hits = np.zeros(2 * n + 1, HIT_DTYPE)
    pool = np.zeros(int(offs[-1]) // 2 + 64 * n + 1024, np.uint32)
    rc = L.orc_cluster_fast_sorted(C.byref(p), _vp(seqs), _vp(offs), n, SORT_MODES[sort], _vp(size_in) if size_in is not None else None,
                                   int(bool(sizein)), _vp(r.seq_unique), _vp(r.uniq_seed), C.byref(nu),
                            _vp(r.uniq_cluster), _vp(r.uniq_nhits), _vp(r.centroid_uniq), _vp(r.cluster_size), C.byref(nc),
                            _vp(hits), len(hits), _vp(pool), len(pool), C.byref(nh), C.byref(cu))
    assert rc == 0, rc
    r.n_unique, r.n_clusters = nu.value, nc.value
    r.uniq_seed = r.uniq_seed[:r.n_unique]; r.uniq_cluster = r.uniq_cluster[:r.n_unique]; r.uniq_nhits = r.uniq_nhits[:r.n_unique]
    r.centroid_uniq = r.centroid_uniq[:r.n_clusters]; r.cluster_size = r.cluster_size[:r.n_clusters]
    r.hits = hits[:nh.value]; r.pool = pool[:cu.value]
    return r


def order_desc_u32(values):
    values = np.ascontiguousarray(values, dtype=np.uint32)
    order = np.zeros(len(values), np.uint32)
    lib().orc_order_desc_u32.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib().orc_order_desc_u32(_vp(values), len(values), _vp(order))
    return order


def cluster_uc_text(r, labels, seqlens, is_nucleo=True):
    """-uc of cluster_fast: per unique in order S / H records followed by its duplicates' H records (outputuc.cpp:10-93),
    then the C records (clustersink.cpp:477-493).  labels / seqlens are per INPUT sequence."""
    L = lib()
    members = [[] for _ in range(r.n_unique)]
    for i, u in enumerate(r.seq_unique):
        members[u].append(i)
    out = []
    buf = C.create_string_buffer(1 << 16)
    hp = 0
    for u in range(r.n_unique):
        seed = int(r.uniq_seed[u]); ql = labels[seed]; qlen = int(seqlens[seed])
        nh = int(r.uniq_nhits[u])
        if nh == 0:
            c = int(r.uniq_cluster[u])
            out.append("S\t%u\t%u\t*\t.\t*\t*\t*\t%s\t*\n" % (c, qlen, ql))
            for m in members[u][1:]:
                out.append("H\t%u\t%u\t100.0\t.\t0\t%u\t=\t%s\t%s\n" % (c, qlen, qlen, labels[m], ql))
        for k in range(nh):
            h = r.hits[hp + k]
            tl = labels[int(r.uniq_seed[int(r.centroid_uniq[int(h["target"])])])]
            for m in members[u]:
                n = L.orc_format_uc_hit(h.ctypes.data if hasattr(h, "ctypes") else r.hits[hp + k:hp + k + 1].ctypes.data,
                                        _vp(r.pool), int(is_nucleo), labels[m].encode(), tl.encode(), buf, len(buf))
                out.append(buf.raw[:n].decode())
        hp += nh
    for c in range(r.n_clusters):
        out.append("C\t%u\t%u\t*\t*\t*\t*\t*\t%s\t*\n" % (c, int(r.cluster_size[c]), labels[int(r.uniq_seed[int(r.centroid_uniq[c])])]))
    return "".join(out)


def strip_size(label):
    """StripSize = StripAnnot(Label, "size=") label.cpp:46-71"""
    if "size=" not in label:
        return label
    fields = label.split(";")                       # Split(myutils.cpp:1588-1607): no field after a trailing separator
    if fields[-1] == "":
        fields.pop()
    new = "".join(f + ";" for f in fields if not f.startswith("size="))
    return new[:-1] if "=" not in new else new


def append_size(label, size):
    """AppendSize -> Psasc (myutils.cpp:824-839)"""
    if label and not label.endswith(";"):
        label += ";"
    return label + "size=%u;" % size


def cluster_centroids_text(r, labels, ss, sizein=False, sizeout=False, minsize=0):
    """-centroids: centroids by decreasing cluster size (QuickSortOrderDesc, clustersink.cpp:262-289), 80-column FASTA;
    labels per MakeCentroidLabel clustersink.cpp:219-243"""
    order = order_desc_u32(r.cluster_size)
    out = []
    for c in order:
        if int(r.cluster_size[int(c)]) < minsize:
            break
        i = int(r.uniq_seed[int(r.centroid_uniq[int(c)])])
        s = ss.seq(i).decode()
        lab = labels[i]
        if sizein or sizeout:
            lab = strip_size(lab)
        if sizeout:
            lab = append_size(lab, int(r.cluster_size[int(c)]))
        out.append(">%s\n" % lab)
        out.extend(s[k:k + 80] + "\n" for k in range(0, len(s), 80))
    return "".join(out)
