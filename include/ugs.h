/*
 * ugs.h - C-ABI of the MI355X-native usearch_global / UCLUST search hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI;
 * its seam is three C++ classes wired by a factory:
 *
 *   class Searcher        /root/reference/src/searcher.h:21-96   (Search(SeqInfo*))
 *   class Aligner         /root/reference/src/aligner.h:23-115   (GlobalAligner::Align)
 *   class HitMgr/HitSink  /root/reference/src/hitmgr.h:16-95, hitsink.h:30-62
 *   MakeDBSearcher()      /root/reference/src/makedbsearcher.cpp:75-236
 *
 * The per-query, synchronous, pointer-chasing interface cannot feed a GPU, so the
 * replacement sits at the same seam but is batched: plain C, caller-owned buffers,
 * integer return codes (0 = ok, <0 = error; never exit()), no C++/torch types.
 * One handle = one GPU = one HIP stream; a handle is used by one host thread at a
 * time; handles on different GPUs are independent.
 *
 * There is NO CPU fallback behind this ABI: every entry point that computes needs a
 * gfx950 device and fails with UGS_E_NODEVICE otherwise.
 */
#ifndef UGS_H
#define UGS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UGS_ABI_VERSION 6   /* 2: accept filters in ugs_params, setup-kernel time in ugs_batch_stats; 3: usearch_local mode
                             * (ugs_params.local..., ugs_hit.raw_score/flags); 4: ugs_db_append, cluster_fast (ugs_cluster_*);
                             * 5: ugs_batch_upload is asynchronous (lifetime rule at its declaration), max_accepts / max_rejects 0 =
                             * unlimited, band 0 = unbanded, ugs_comm.h (RCCL gather); 6: ugs_device_synchronize, ugs_debug_alloc_stats
                             * (additive) */

/* error codes */
#define UGS_OK            0
#define UGS_E_ARG        -1   /* bad argument / unsupported option value            */
#define UGS_E_NODEVICE   -2   /* no usable gfx950 device (there is no CPU fallback)  */
#define UGS_E_HIP        -3   /* HIP runtime error (see ugs_last_error)             */
#define UGS_E_NOMEM      -4
#define UGS_E_CAPACITY   -5   /* caller-provided output buffer too small             */
#define UGS_E_ENVELOPE   -6   /* input outside the device path's supported envelope  */

/*
 * Snapshot of every option that reaches the hot path.  The reference
 * reads these ad hoc deep in hot code (oget_*); here they are frozen once.
 *   id            -id as the reference stores it for ranking: float  (udbusortedsearcher.cpp:101)
 *   id_accept     -id as the accepter compares it: the option table stores floats
 *                 (opts.cpp:265), so this is (double)(float)id       (accepter.cpp:35-39)
 *   max_accepts / max_rejects                                         (terminator.cpp:8-45)
 *   big           -big: DB size above which the "Big" ranker is used  (udbusortedsearcher.cpp:44)
 *   bump_pct      -bump                                               (udbusortedsearcher.cpp:278)
 *   stepwords     -stepwords                                          (wordparams.cpp:179-191)
 *   band, minhsp, xdrop_nw, hsp_word_len                              (alnheuristics.cpp:26-62)
 *   match, mismatch (nt) ; aa uses BLOSUM62                           (alnparams.cpp:333,380-384)
 *   dbmask        0 = upper-case only, 1 = fastnucleo/fastamino       (makeudb.cpp:11-25);
 *                 3 = fastnucleo/fastamino with -hardmask ('N' / 'X' instead of lower case, fastmask.cpp:98,117-150);
 *                 2 = letters are used as given: the stored, already masked letters of a .udb (loaddb.cpp:100-125)
 *   filter_mask + values: the optional accept filters of Accepter::IsAcceptLo (accepter.cpp:41-91): -maxid (only
 *                 tested when -id is set), -mincols, -maxgaps, -query_cov, -max_query_cov, -target_cov,
 *                 -max_target_cov, -maxdiffs, -mindiffs.  A filter is active when its UGS_F_* bit is set; float
 *                 values are compared as (double)(float)value like every option (opts.cpp:265).  A hit that fails
 *                 one is a reject for the terminator, exactly like a failed -id.  UGS_F_ABSKEW (-abskew: target size /
 *                 query size from the ;size= annotations, arscorer.cpp:809-816) needs the pair keys below.
 *   pair_mask + values: the pair filters of Accepter::RejectPair (accepter.cpp:140-197): -self (equal labels), -notself,
 *                 -selfid (same length and identical stored letters), -min_sizeratio, -minqt/-maxqt (query length /
 *                 target length), -minsl/-maxsl (shorter / longer).  usearch_global only.  On the Big ranking path a
 *                 rejected pair is a reject for the terminator (udbusortedsearcherbig.cpp:118-127); on the small path it is
 *                 skipped without being counted (udbusortedsearcher.cpp:138-151 ignores SetTarget's result, the aligner
 *                 then refuses the pair, searcher.cpp:63-67): there refused targets are dropped where the candidates are
 *                 chosen (k_rank), except for -selfid, which needs the letters: the device keeps up to 32 candidates more
 *                 per strand for it and ugs_batch_sync fails with UGS_E_ENVELOPE if a walk passes over more identical
 *                 sequences than that (never a silently shortened walk).  Labels and sizes reach the
 *                 device as integer keys: ugs_db_set_pair_keys / ugs_batch_set_pair_keys.
 *   local         0 = usearch_global; 1 = usearch_local: the same U-sort candidate walk, but every candidate goes
 *                 through LocalAligner2::AlignMulti (localmulti.cpp:9-118: seed every hsp_word_len-mer the target
 *                 shares with the query, LocalAligner::AlignPos localaligner.cpp:101-222: ungapped x-drop xdrop_u,
 *                 anchor, gapped x-drop xdrop_g, e-value gate) and may yield several HSPs.  id may be left unset
 *                 (ranking then uses 0.5, makedbsearcher.cpp:172: oget_fltd(OPT_id, 0.5)).
 *   evalue        -evalue (required by usearch_local), compared as (double)(float)
 *   ka_dbsize     -ka_dbsize; its default 1e9 counts as "filled" (o_defaults.inc:2, opts.cpp:187-192), so the DB
 *                 letter count is never used (makedbsearcher.cpp:89-95)
 *   local_open/local_ext  -lopen/-lext as penalties: -10 / -1 for both alphabets (alnparams.cpp:362-369)
 *   max_hsps      hit slots per (query strand, accepted target); more HSPs than that => UGS_E_CAPACITY
 */
/* pair filters of Accepter::RejectPair (accepter.cpp:140-197), ugs_params.pair_mask */
enum {
  UGS_P_SELF = 1, UGS_P_NOTSELF = 2, UGS_P_SELFID = 4, UGS_P_MIN_SIZERATIO = 8, UGS_P_MINQT = 16, UGS_P_MAXQT = 32,
  UGS_P_MINSL = 64, UGS_P_MAXSL = 128
};
enum {
  UGS_F_MAXID = 1, UGS_F_MINCOLS = 2, UGS_F_MAXGAPS = 4, UGS_F_QUERY_COV = 8, UGS_F_MAX_QUERY_COV = 16,
  UGS_F_TARGET_COV = 32, UGS_F_MAX_TARGET_COV = 64, UGS_F_MAXDIFFS = 128, UGS_F_MINDIFFS = 256, UGS_F_ABSKEW = 512
};
typedef struct ugs_params {
  int32_t  is_nucleo;
  int32_t  word_len;       /* UDB word length: 8 nt / 5 aa            */
  float    id;
  double   id_accept;
  int32_t  id_set;         /* 0 = -id absent: no identity filter (accepter.cpp:35 tests ofilled) */
  int32_t  strand_both;
  int32_t  max_accepts;
  int32_t  max_rejects;
  uint32_t big;
  uint32_t bump_pct;
  uint32_t stepwords;
  int32_t  band;
  int32_t  minhsp;
  float    xdrop_nw;
  float    match;
  float    mismatch;
  int32_t  hsp_word_len;   /* 5 nt / 3 aa                              */
  int32_t  dbmask;
  uint32_t filter_mask;    /* UGS_F_* bits                             */
  float    maxid, query_cov, max_query_cov, target_cov, max_target_cov;
  uint32_t mincols, maxgaps, maxdiffs, mindiffs;
  int32_t  local;
  float    evalue;
  float    xdrop_u;        /* 16 (o_defaults.inc:22)                   */
  float    xdrop_g;        /* 32 (o_defaults.inc:20)                   */
  float    local_open;     /* -10                                      */
  float    local_ext;      /* -1                                       */
  float    ka_dbsize;      /* 1e9                                      */
  uint32_t max_hsps;       /* 8                                        */
  uint32_t pair_mask;      /* UGS_P_* bits                             */
  float    min_sizeratio, minqt, maxqt, minsl, maxsl, abskew;
  uint32_t align_flags;    /* UGS_A_FULLDP | UGS_A_GAFORCE | UGS_A_TERMID | UGS_A_TERMIDD */
  float    termid, termidd;
} ugs_params;
/* -fulldp: no HSPs, one unbanded Viterbi over the whole pair (globalalignmem.cpp:148-152, ViterbiFastMem);
 * -gaforce: a pair without good HSPs is aligned all the same (FailIfNoHSPs = false, globalaligner.cpp:9-12) */
/* -termid / -termidd (terminator.cpp:66-87, usearch_global only): the walk of a query - both strands, they share the HitMgr -
 * also ends once its worst hit is at or below termid, or its best and worst hits are more than termidd apart */
enum { UGS_A_FULLDP = 1, UGS_A_GAFORCE = 2, UGS_A_TERMID = 4, UGS_A_TERMIDD = 8 };

/*
 * One accepted hit == one AlignResult appended to HitMgr (hitmgr.cpp:161-183), with
 * the fields AlignResult::FillLo derives from the path (arscorer.cpp:201-296).
 * Coordinates are 0-based positions of the first/last aligned (M) column.
 * The alignment path is run-length encoded in the cigar pool: one uint32 per run,
 * (length << 2) | op with op 0=M 1=D 2=I, in alignment order INCLUDING terminal gaps
 * (what CompressPath prints into .uc, comppath.cpp:7-48).
 */
typedef struct ugs_hit {
  uint32_t query;        /* index into the batch                           */
  uint32_t target;       /* DB sequence index (uc column 2)                */
  uint32_t ids;          /* m_IdCount                                      */
  uint32_t mism;         /* m_MismatchCount                                */
  uint32_t gaps_int;     /* m_IntGapCount (gap columns between first/last M) */
  uint32_t aln_len;      /* m_AlnLength  (terminal gaps excluded)          */
  uint32_t opens;        /* GetGapOpenCount (arscorer.cpp:554-569)         */
  uint32_t qlo, qhi, tlo, thi;
  uint32_t ql, tl;       /* sequence lengths                               */
  uint32_t strand;       /* 0 = plus, 1 = query reverse-complemented       */
  uint64_t cigar_off;    /* into the cigar pool, in uint32 units           */
  uint32_t cigar_len;    /* number of runs                                 */
  uint32_t cols;         /* total path columns incl. terminal gaps         */
  float    raw_score;    /* local hits: HSP raw score (AlignResult::GetRawScore arscorer.cpp:87-103); 0 for global */
  uint32_t flags;        /* UGS_HIT_LOCAL: a usearch_local HSP - qlo..thi are the HSP (m_HSP), the path covers only it */
} ugs_hit;
#define UGS_HIT_LOCAL 1u
/* flags bits 8..31: the hit's position in HitMgr's append order within its query (candidate order, plus strand first);
 * HitMgr::GetTopHit (hitmgr.cpp:398-415) keeps the earlier of two hits with equal score and equal target */
#define UGS_HIT_ORDER_SHIFT 8

typedef struct ugs_db ugs_db;       /* opaque: masked DB + UDB index resident in HBM */
typedef struct ugs_batch ugs_batch; /* opaque: one query batch resident in HBM       */

/* Fill *p with the reference defaults for usearch_global (o_defaults.inc, terminator.cpp:26-31). */
int ugs_params_init(ugs_params *p, int is_nucleo, double id);
/* Switch *p to usearch_local (cmd_usearch_local searchcmd.cpp:42-45, makedbsearcher.cpp:87-121): -evalue is required;
 * id_set = 0 means no -id on the command line (no identity filter, ranking with 0.5). */
int ugs_params_set_local(ugs_params *p, double evalue, int id_set);
/* Karlin-Altschul numbers of a local hit (EStats::RawScoreToBitScore / RawScoreToEvalue estats.cpp:72-96 with the
 * BLAST gapped constants and -ka_dbsize), evaluated exactly as the reference binary does (see ugs_host.cpp). */
int ugs_local_evalue(const ugs_params *p, double raw_score, uint32_t ql, double *evalue, double *bits);

int ugs_abi_version(void);
int ugs_device_count(void);
/* Blocks until everything enqueued on `device` by this process has finished (hipDeviceSynchronize): what a multi-rank driver puts
 * beside its inter-process barrier on both sides of a timed region (the reference has no counterpart: search.cpp:121-128 joins threads). */
int ugs_device_synchronize(int device);

/*
 * Replaces LoadUDB + UDBData::FromSeqDB (loaddb.cpp:100-125, udbbuild.cpp:303-398):
 * `seqs` are the raw DB letters (as read from FASTA, concatenated, no terminators),
 * offs[nseq+1] their boundaries.  Masking (fastmask.cpp:88-158) and the word index are
 * built here and stay resident on `device`.
 */
int ugs_db_create(const ugs_params *p, const char *seqs, const uint64_t *offs,
                  uint32_t nseq, int device, ugs_db **out);
void ugs_db_destroy(ugs_db *db);
/* introspection used by tests/bench: number of index postings, slots, bytes in HBM */
/* Per-sequence keys for the pair filters and -abskew: label_key[i] identifies sequence i's label (equal labels <=> equal
 * keys, the caller interns the strings; the same key space for DB and queries), size[i] is its ;size= annotation or
 * UINT32_MAX when it has none (GetSizeFromLabel(label, UINT_MAX) accepter.cpp:150-151).  Either array may be NULL when no
 * active filter needs it.  Required before a search whenever pair_mask or UGS_F_ABSKEW is set. */
int ugs_db_set_pair_keys(ugs_db *db, const uint32_t *label_key, const uint32_t *size);
int ugs_batch_set_pair_keys(ugs_batch *b, const uint32_t *label_key, const uint32_t *size);   /* after ugs_batch_upload */

int ugs_db_stats(const ugs_db *db, uint64_t *n_postings, uint64_t *n_slots, uint64_t *hbm_bytes);

/*
 * Replaces the Thread() loop calling Searcher::Search per query (search.cpp:51-87,
 * searcher.cpp:122-161): search a whole batch; hits come back grouped by query in
 * query order, each group sorted as HitMgr::Sort orders them (hitmgr.cpp:477-483).
 * nhits_per_query[nq] receives the group sizes.  Returns UGS_E_CAPACITY (cigar_used = runs needed) if hits_cap or
 * cigar_cap (uint32 units) is too small (hits_cap = nq * max_accepts * (1+strand_both)
 * always suffices; max_accepts 0 = unlimited: a query may have as many hits as it has candidates - grow hits_cap and call again).
 * Walk depth: max_accepts + max_rejects - 1 may exceed the 64 candidates a ranking pass keeps per strand, and either may be 0 =
 * unlimited, as in the reference (terminator.cpp:22-31,64-100): the walks that use up their 64 candidates are continued over the
 * query's complete sorted candidate list (usearch_global and usearch_local; -termid / -termidd are refused together with such settings).
 */
int ugs_search_batch(ugs_db *db, const char *qseqs, const uint64_t *qoffs, uint32_t nq,
                     ugs_hit *hits, uint64_t hits_cap, uint32_t *nhits_per_query,
                     uint32_t *cigar_pool, uint64_t cigar_cap, uint64_t *cigar_used);

/*
 * Staged form of the same call, for callers that keep batches resident in HBM
 * (bench.py times ugs_batch_search + ugs_batch_sync only):
 *   upload  = H2D of letters/offsets
 *   search  = enqueue the ranking + alignment kernels on the handle's stream
 *   sync    = wait for the stream
 *   fetch   = D2H of hit records + cigar runs, grouped/sorted as ugs_search_batch
 */
int ugs_batch_create(ugs_db *db, uint32_t max_queries, uint64_t max_letters, ugs_batch **out);
void ugs_batch_destroy(ugs_batch *b);
/* ugs_batch_upload is ASYNCHRONOUS since ABI 5: it validates, enqueues the H2D copies on the batch's own copy stream and
 * returns; ugs_batch_search orders its kernels behind them with an event.  Rules for the caller:
 *   - qseqs[qoffs[0] .. qoffs[nq]) must stay allocated and unmodified until the next ugs_batch_sync (or ugs_batch_destroy)
 *     of this batch returns; qoffs itself is consumed before the call returns.  From page-locked memory
 *     (ugs_host_register) the copy overlaps other batches' kernels; from pageable memory the runtime stages it.
 *   - uploading into a batch whose previous search has not been synced is allowed: the copies queue behind that
 *     search on the device (its inputs are not overwritten under it), but its RESULTS are then lost - fetch first.
 * ugs_search_batch (the one-call form above) keeps the synchronous contract: it returns with everything consumed. */
int ugs_batch_upload(ugs_batch *b, const char *qseqs, const uint64_t *qoffs, uint32_t nq);
/* Blocks until the letters and offsets of the last ugs_batch_upload have arrived in HBM (a caller that wants to time the search alone,
 * or to reuse a non-pinned source buffer early; ugs_batch_search itself waits on the device, not on the host). */
int ugs_batch_wait_upload(ugs_batch *b);
int ugs_batch_search(ugs_batch *b);
int ugs_batch_sync(ugs_batch *b);
int ugs_batch_fetch(ugs_batch *b, ugs_hit *hits, uint64_t hits_cap, uint32_t *nhits_per_query,
                    uint32_t *cigar_pool, uint64_t cigar_cap, uint64_t *cigar_used);
/*
 * Per-stage device time of the last ugs_batch_search (HIP events on the handle's own
 * stream), and the algorithmic work it did: postings the reference
 * semantics requires reading, letters of query + aligned candidates, DP cells.
 */
typedef struct ugs_batch_stats {
  float    ms_rank;          /* ranking kernel (k_rank) alone             */
  float    ms_align;         /* alignment kernel                          */
  float    ms_total;
  uint64_t postings;         /* sum over queries of P(q)                  */
  uint64_t query_letters;
  uint64_t target_letters;   /* bytes of candidate targets fetched by the alignment stage: the letters, or for nt targets up to 1024
                              * letters their 2-bit planes (8 bytes per 16 letters) + the letters of the pairs that reach the chain gate */
  uint64_t pairs_aligned;
  uint64_t dp_cells;
  uint64_t hits;
  float    ms_rank_setup;    /* sampled-row selection kernel (k_rank_setup), runs before k_rank */
  float    reserved_;
} ugs_batch_stats;
int ugs_batch_get_stats(ugs_batch *b, ugs_batch_stats *st);

/* HitMgr::Sort (hitmgr.cpp:477-483) on a hit table grouped by query, in place: the order ugs_batch_fetch returns.
 * For tables taken from ugs_batch_device_results (multi-GPU gather), which are in candidate order.  Host only. */
int ugs_hits_sort(ugs_hit *hits, const uint32_t *nhits_per_query, uint32_t nq, int local);

/*
 * Stage-level entry point used by the parity tests: the ranked candidate list of every
 * query exactly as the reference's candidate loop would walk it
 * (udbusortedsearcherbig.cpp:113-134 / udbusortedsearcher.cpp:138-151), truncated to
 * the first k = min(64, max_accepts + max_rejects - 1) entries per strand.
 * cand[(q*nstrand + s)*k + j] = target index, cnt[...] = word count, n[q*nstrand+s] = entries.
 */
int ugs_batch_get_candidates(ugs_batch *b, uint32_t *cand, uint32_t *cnt, uint32_t *n, uint32_t k_cap);
/* k of this batch: max_accepts + max_rejects - 1, plus the spare candidates kept for -selfid on the small path (<= 64; deeper walks: 64) */
int ugs_batch_candidate_k(const ugs_batch *b, uint32_t *k);
/* Diagnostic: which ranking code the last synced search of this batch ran (the test-suite asserts that every compiled path is
 * reached by an oracle-compared test).  out[0] = units ranked by the bitmap kernel (ugs_rank2.hip), out[1] = units it deferred to the
 * general kernel, out[2] = the general kernel's instantiation (big | counter bits << 1 | fast8 << 8 | longrows << 9 | wide offsets << 10), out[3] = 1 if
 * the bitmap kernel was launched; with n >= 6 also out[4] / out[5] = microseconds of the bitmap kernel / of the general kernel behind it
 * (HIP events on the handle's stream); with n >= 7 also out[6] = candidate pairs k_align rejected through its group filter (pairs without
 * an HSP, tested four at a time); with n >= 8 also out[7] = WHICH kernel of the bitmap family ran: 0 none, 1 k_rank2, 2 k_rank2g, 3 k_rank3g
 * (sparse index, two filter passes), 4 k_rank2's cluster_fast instantiation, 5 k_rank2 over 16-bit postings; with n >= 9 also out[8] = units of
 * the deferred list that the heavy-unit instantiation ranked (cluster_fast; the rest of the list went on to the general kernel).  n >= 4. */
int ugs_debug_kernel_hits(const ugs_batch *b, uint64_t *out, int n);
/* Diagnostic: *seen = the ranking kernels this process has launched so far, *compiled = the ones the library holds, one bit each
 * (bits 0-4: Big path - 4-bit counters, its long-row twin, 8/16-bit flattened (sparse index), 8/16-bit dense, 8/16-bit dense + long
 * rows; bits 5-9: the same five on the small path; 12 / 13: the two Big-path 4-bit kernels with 64-bit offsets; 14: the bitmap kernel,
 * 15: its gather variant for sparse indexes, 16: its cluster_fast instantiation, 17: k_rank3g, the two-pass filter kernel for sparse indexes,
 * 18: the bitmap kernel over 16-bit partition-relative postings, 19: its heavy-unit instantiation for cluster_fast).  The test-suite ends with seen == compiled. */
int ugs_debug_rank_instances(uint64_t *seen, uint64_t *compiled);
/* Diagnostic: the name of the ranking kernel behind bit `bit` of those masks (a static string), or NULL if the library holds no such
 * instantiation - the library's own table, so that a test or tool keeps no list of its own. */
const char *ugs_debug_rank_instance_name(int bit);
/* Diagnostic: the deep-walk stage of the last synced search of this batch (walks that need more than the 64 candidates a ranking
 * pass keeps - max_accepts + max_rejects - 1 > 64, or 0 = unlimited: terminator.cpp:22-31,64-100 has no depth limit): how many walks
 * were parked and continued over their complete sorted candidate list, and how many keys those lists held. */
int ugs_debug_deep_walks(const ugs_batch *b, uint64_t *parked_units, uint64_t *list_keys);
/* Diagnostic: the library's device allocator.  out[0] = UGS_GUARD_ALLOC mode (0 = hipMalloc / hipFree; 1 / 2 = every buffer a mapping of
 * its own, right-aligned against an unmapped page: an out-of-bounds access of any kernel faults deterministically), out[1] / out[2] =
 * guarded allocations made / released, out[3] / out[4] = bytes mapped now / at the peak. */
int ugs_debug_alloc_stats(unsigned long long out[5]);

/*
 * Debug / tuning switches.  NOT part of the contract: they exist for A/B measurements and fault isolation, are read from the
 * environment ONCE per database handle (at ugs_db_create, never inside a search call) and default to "unset":
 *   UGS_QPK=1               nt query letters are packed once per unit by the setup kernel (2 bits + other-letter plane) for k_align
 *   UGS_ALIGN_GROUP=n       k_align tests the candidates of a unit four at a time once n of them were rejected (default 1; 0 = never)
 *   UGS_NO_PACKED=1         k_align fetches every target from the byte array instead of the packed letters
 *   UGS_LONGROWS=0|1        force the long-row ranking instantiations off / on
 *   UGS_GSIZE=n UGS_GSHIFT=k  partition size of k_rank (targets, a multiple of 64 / a power of two)
 *   UGS_RANK_WGS_PER_CU=n UGS_ALIGN_WGS_PER_CU=n   cap on resident workgroups per CU
 *   UGS_EMIT_LIMIT=n        candidate-key buffer of k_rank (forces the regrow path)
 *   UGS_WIDE_OFFSETS=1      k_rank's Big-path 4-bit kernels with 64-bit table / row offsets (chosen by themselves for an index that needs them)
 *   UGS_RANK2=0|1           bitmap ranking kernel off / on wherever the index allows it (default: on for dense Big-path indexes)
 *   UGS_R2_G=n UGS_R2_KCAP=n UGS_R2_WAVES=n   its partition size (multiple of 8192), kept-key capacity, waves per CU
 *   UGS_SETUP_STREAM=1      the unit set-up kernel of a search runs on a second stream, beside the kernels of the batch enqueued in front (A/B)
 *   UGS_R2_HV=0             cluster_fast: the units the bitmap kernel defers go straight to the general kernel (no heavy-unit stage; A/B); 2: EVERY unit is deferred to the heavy-unit stage (tests)
 *   UGS_R2_P16=0            the bitmap kernel streams the 32-bit postings instead of the 16-bit partition-relative copy (A/B)
 *   UGS_R3=0|1 UGS_R3_SP=n UGS_R3_PPS=n       sparse (protein) Big-path index: 0 = k_rank2g instead of k_rank3g; k_rank3g's partitions per
 *                           super-partition (default: per unit, so that a super-partition holds ~ UGS_R3_PPS = 4096 of its postings)
 *   UGS_DEBUG_SYNC=1 UGS_PHASE_CLOCKS=1       finish and log every stage / print the kernels' phase clocks with the stats
 * Read once per PROCESS (ugs_alloc.cpp):
 *   UGS_GUARD_ALLOC=1|2 UGS_GUARD_ALIGN=n     the guard allocator (ugs_debug_alloc_stats; 2: freed address ranges stay reserved; n: alignment of the
 *                           pointers handed out, default 16 bytes)
 *   UGS_ABORT_BT=file|1     a SIGABRT handler that writes the backtrace of the aborting THREAD (a runtime thread reporting a GPU fault, glibc
 *                           reporting a corrupted heap) to `file` (1: stderr) and chains to the previous handler
 * (ugs_cluster_fast reads UGS_CLUSTER_BATCH / UGS_CLUSTER_PROFILE at its start; ugs_cli reads UGS_CLI_PROFILE / UGS_CLI_FORCE_GATHER.)
 */

/*
 * Device-resident, query-grouped results of the last synced search (valid until the next search
 * or destroy): compact hits[n] (ugs_hit records with `query` offset by query_base; hits of one
 * query adjacent, strand 0 first, discovery order), nhits_per_query[nq] (uint32) and the run pool.
 * For callers that move results GPU-to-GPU (the multi-GPU driver gathers them to rank 0 with RCCL
 * over xGMI) instead of fetching to the host.
 */
/* A shard's offset into the global query numbering (multi-GPU: rank r's first query).  Set before ugs_batch_search, the
 * search's own grouping stamps ugs_hit.query with it and ugs_batch_device_results(query_base = the same value) then hands the
 * table over as it stands, with no regrouping pass between the search and the gather.  ugs_batch_fetch always returns
 * batch-local query indexes.  Default 0. */
int ugs_batch_set_query_base(ugs_batch *b, uint32_t query_base);
int ugs_batch_device_results(ugs_batch *b, uint32_t query_base, void **d_hits, uint64_t *hits_bytes,
                             void **d_nhits, uint64_t *nhits_bytes, void **d_cigar, uint64_t *cigar_bytes);

/*
 * Text writers replacing OutputSink::OutputBlast6 / OutputUC / OutputUCNoHits
 * (blast6out.cpp:27-80, outputuc.cpp:10-93).  Write at most cap bytes (incl. NUL) of one
 * line (with trailing '\n') into buf; return the length that was / would be written.
 */
int ugs_format_blast6(const ugs_hit *h, const char *qlabel, const char *tlabel, char *buf, int cap);
/* -blast6out line of a usearch_local hit (blast6out.cpp:27-80, local branch :71-77) */
int ugs_format_blast6_local(const ugs_params *p, const ugs_hit *h, const char *qlabel, const char *tlabel, char *buf, int cap);
int ugs_format_uc_hit(const ugs_hit *h, const uint32_t *cigar_pool, int is_nucleo,
                      const char *qlabel, const char *tlabel, char *buf, int cap);
int ugs_format_uc_nohit(uint32_t ql, const char *qlabel, char *buf, int cap);

/* ------------------------------------------------------------------------------------------
 * Gapped x-drop extension: the kernel the reference's local aligner
 * calls per anchor (localaligner.cpp:190, :330).  Batched: every job is one call of
 *   mode UGS_XDROP_ALIGN : XDropAlignMem   xdropalignmem.cpp:217-244 (-> :26-214; sides longer than
 *                          g_MaxL = 4096 go through XDropFwdSplit / XDropBwdSplit
 *                          xdropfwdsplit.cpp:24-91, xdropbwdsplit.cpp:15-79)
 *   mode UGS_XDROP_FWD   : XDropFwdFastMem xdropfwdmem.cpp:344-749 on (A,B) from their first letters
 *   mode UGS_XDROP_BWD   : XDropBwdFastMem xdropbwdmem.cpp:23-70   on (A,B) from their last letters
 * (FWD/BWD require LA,LB <= 4096 as the reference's callers guarantee.)
 * Scores are exact small integers held in float by the reference (alnparams.cpp:362-369:
 * LocalOpen/LocalExt = -lopen/-lext = -10/-1 for both alphabets - the option defaults count as
 * "filled", opts.cpp:187-192, so the aa -5 branch at :373-376 never runs; substitution matrix as
 * for the global path).
 */
enum { UGS_XDROP_ALIGN = 0, UGS_XDROP_FWD = 1, UGS_XDROP_BWD = 2 };

typedef struct ugs_xdrop_params {
  int32_t is_nucleo;
  float   match, mismatch;      /* nt matrix (setnucmx.cpp:11-99); aa: BLOSUM62 implied            */
  float   local_open, local_ext;/* negative penalties, AlnParams::GetLocalOpen/Ext                  */
  float   xdrop;                /* X: -xdrop_g, default 32 (o_defaults.inc:20)                      */
} ugs_xdrop_params;

typedef struct ugs_xdrop_job {
  uint32_t a, b;                /* sequence indices into the A set / B set                          */
  uint32_t anc_loi, anc_loj, anc_len;   /* anchor (ALIGN mode only)                                 */
  uint32_t mode;
} ugs_xdrop_job;

typedef struct ugs_xdrop_hsp {  /* HSPData (hsp.h:4-11) + where the path went                        */
  float    score;               /* 0 => no alignment (path empty)                                    */
  uint32_t loi, loj, leni, lenj;
  uint32_t path_len;            /* runs in the pool: len<<2 | op, op 0=M 1=D 2=I                     */
  uint64_t path_off;
} ugs_xdrop_hsp;

void ugs_xdrop_params_init(ugs_xdrop_params *p, int is_nucleo);
/* host buffers in, host buffers out; results in job order; path_pool filled in job order.
 * UGS_E_CAPACITY if path_cap is too small (path_used then holds the needed size). */
int ugs_xdrop_batch(int device, const ugs_xdrop_params *p,
                    const char *a_seqs, const uint64_t *a_offs, uint32_t na,
                    const char *b_seqs, const uint64_t *b_offs, uint32_t nb,
                    const ugs_xdrop_job *jobs, uint32_t njobs,
                    ugs_xdrop_hsp *hsps, uint32_t *path_pool, uint64_t path_cap, uint64_t *path_used);
/* device time of the last ugs_xdrop_batch on this thread (kernel only, HIP events) and its DP cells */
int ugs_xdrop_last_stats(float *ms_kernel, uint64_t *dp_cells);

/* ------------------------------------------------------------------------------------------
 * .udb files: the reference's on-disk database, written by -makeudb_usearch
 * (UDBData::ToUDBFile udbio.cpp:280-352 + SeqDB::ToFile seqdbio.cpp:17-113) and read by LoadUDB
 * (loaddb.cpp:100-125 -> UDBData::FromUDBFile udbio.cpp:242-278 + SeqDB::FromFile seqdbio.cpp:160-235).
 * Layout: packed UDBFileHdr (udbfile.h:18-49, 200 bytes) | uint32 row sizes[slots] | 'UDB3' | rows
 * (uint32 target indexes, ascending) | 'UDB4' | SeqDBFileHdr (32 bytes) | label offsets | labels |
 * lengths | letters (as masked at makeudb time).
 * Only the default index flavour is supported (unhashed, unspaced, uncoded, dbstep 1, dbaccel 100);
 * anything else returns UGS_E_ENVELOPE.
 */
typedef struct ugs_udb_info {
  int32_t  is_nucleo;
  uint32_t word_len;
  uint64_t nseq, nletters, label_bytes;   /* labels: NUL-terminated, concatenated in target order */
  uint64_t slots, n_postings;
} ugs_udb_info;
int ugs_udb_stat(const char *path, ugs_udb_info *info);
/* Any of the output pointers may be NULL.  seqs[nletters], offs[nseq+1], labels[label_bytes],
 * row_sizes[slots], postings[n_postings] (rows concatenated in slot order). */
int ugs_udb_read(const char *path, char *seqs, uint64_t *offs, char *labels, uint32_t *row_sizes, uint32_t *postings);
/* Writes the database held by `db` (masked letters + the index built on the GPU) in the reference's format;
 * byte-identical to the reference's -makeudb_usearch output for the same FASTA.  labels: nseq NUL-terminated strings. */
int ugs_udb_write(const char *path, const ugs_db *db, const char *labels, uint64_t label_bytes);

/* ------------------------------------------------------------------------------------------
 * Output fidelity beyond blast6/uc; host-side formatting of device results.
 *
 * ugs_format_userout: OutputSink::OutputUser (h != NULL) / OutputUserNoHits (h == NULL), userout.cpp:47-352.
 * fields = the -userfields string ("query+target+id+..."); supported names: query target clusternr evalue id
 * fractid dist mid pctpv pctgaps pairs gaps allgaps qlo qhi tlo thi qlor qhir tlor thir qlot qhit qunt tlot thit
 * tunt pv ql tl qs ts alnlen opens exts raw bits aln caln qseq tseq qseg tseg qstrand tstrand qrow trow qrowdots
 * trowdots qframe tframe orflo orfhi orfframe mism ids qcov tcov diffs diffsa editdiffs (global-alignment
 * semantics, AlignResult getters arscorer.cpp / alignresult.h:97-240); anything else -> UGS_E_ARG.
 * qseq is the query as read (it is reverse-complemented here for a minus-strand hit); tseq the target as the
 * reference holds it (i.e. the masked DB letters for -db x.fa / x.udb).
 * Like the other writers: returns the line length (excluding NUL), or a negative error code.
 */
int ugs_userfields_check(const char *fields);
int ugs_format_userout(const ugs_hit *h, const uint32_t *cigar_pool, int is_nucleo, const char *fields,
                       const char *qlabel, const char *tlabel, const char *qseq, uint32_t ql,
                       const char *tseq, uint32_t tl, char *buf, int cap);
/* -alnout: ugs_format_alnout_header = OutputSink::OutputReport (outputsink.cpp:243-258,338-356: "Query >label" and the
 * %Id / TLen / Target table of the query's reported hits; empty for a query without hits), then one
 * ugs_format_alnout_hit = WriteAln (alnout.cpp:41-171) per hit: rows of 80 columns with 1-based position labels,
 * the annotation row (| identical, + IUPAC match / : . BLOSUM62 >= 2 / > 0) and the summary line. */
/* -userout line of a usearch_local hit: HSP coordinates / segments / coverages, real evalue, raw, bits (userout.cpp:148-207
 * over the local branches of the AlignResult getters) */
int ugs_format_userout_local(const ugs_params *p, const ugs_hit *h, const uint32_t *cigar_pool, const char *fields,
                             const char *qlabel, const char *tlabel, const char *qseq, uint32_t ql,
                             const char *tseq, uint32_t tl, char *buf, int cap);
/* -alnout for usearch_local hits: OutputReportLocal outputsink.cpp:260-298, WriteAln's local summary alnout.cpp:151-163 */
int ugs_format_alnout_header_local(const ugs_params *p, const ugs_hit *hits, uint32_t n, const char *qlabel, const char *const *tlabels, char *buf, int cap);
int ugs_format_alnout_hit_local(const ugs_params *p, const ugs_hit *h, const uint32_t *cigar_pool, const char *qlabel, const char *tlabel,
                                const char *qseq, uint32_t ql, const char *tseq, uint32_t tl, char *buf, int cap);
int ugs_format_alnout_header(const ugs_hit *hits, uint32_t n, const char *qlabel, const char *const *tlabels, char *buf, int cap);
int ugs_format_alnout_hit(const ugs_hit *h, const uint32_t *cigar_pool, int is_nucleo, const char *qlabel, const char *tlabel,
                          const char *qseq, uint32_t ql, const char *tseq, uint32_t tl, char *buf, int cap);

/* -fastapairs (OutputFastaPairs outputsink.cpp:231-241) and -qsegout / -tsegout (OutputQSeg/OutputTSeg :203-229,
 * RowToFasta :30-55; which = 0 query, 1 target): the aligned rows of a hit as FASTA */
int ugs_format_fastapairs(const ugs_hit *h, const uint32_t *cigar_pool, const char *qlabel, const char *tlabel,
                          const char *qseq, uint32_t ql, const char *tseq, uint32_t tl, char *buf, int cap);
int ugs_format_segout(const ugs_hit *h, const uint32_t *cigar_pool, int which, const char *qlabel, const char *tlabel,
                      const char *qseq, uint32_t ql, const char *tseq, uint32_t tl, char *buf, int cap);
/* OutputBlast6NoHits blast6out.cpp:82-103 (-output_no_hits) */
/* -trimout record (OutputTrim outputsink.cpp:401-415, GetTrimInfo arscorer.cpp:933-971): the query minus its overhang */
int ugs_format_trimout(const ugs_hit *h, const uint32_t *cigar_pool, const char *qlabel, const char *qseq, uint32_t ql, char *buf, int cap);
int ugs_format_blast6_nohit(const char *qlabel, char *buf, int cap);
/* SeqToFasta seqdb.cpp:62-90 (-matched / -notmatched / -dbmatched / -dbnotmatched records) */
int ugs_format_fasta(const char *label, const char *seq, uint32_t len, char *buf, int cap);
/* HitMgr::GetHitCount / GetHit hitmgr.cpp:366-393,464-475: how many of one query's (sorted) hits are reported under
 * -maxhits (0 = unset) / -top_hit_only / -top_hits_only, starting at hits[*first] (0 unless -top_hit_only, where
 * the single reported hit is GetTopHit: best score, ties to the smallest target index) */
uint32_t ugs_hits_to_report(const ugs_hit *hits, uint32_t n, uint32_t maxhits, int top_hit_only, int top_hits_only,
                            uint32_t *first);
/* the DB letters as the reference holds them after MaskDB (makeudb.cpp:11-25): out[nletters] */
int ugs_db_masked_letters(const ugs_db *db, char *out);

/* ------------------------------------------------------------------------------------------
 * otutab sink (cmd_otutab searchcmd.cpp:20-40: usearch_global with -id 0.97 -maxaccepts 3 -maxrejects 32
 * -stepwords 0 -strand both, hits consumed by OTUTableSink otutabsink.cpp:31-58).  Feed every query in input order
 * with the label of its top hit (ugs_hits_to_report(..., top_hit_only=1, ...) = HitMgr::GetTopHit) or NULL;
 * map_line receives the -mapout line ("query<TAB>otu\n", empty when unassigned).  ugs_otutab_write = -otutabout
 * (OTUTable::ToTabbedFile otutab.cpp:247-313: rows / columns in order of first appearance).
 */
typedef struct ugs_otutab ugs_otutab;
ugs_otutab *ugs_otutab_create(void);
void ugs_otutab_destroy(ugs_otutab *t);
int ugs_otutab_add(ugs_otutab *t, const char *qlabel, const char *top_hit_tlabel, char *map_line, int cap);
int ugs_otutab_write(const ugs_otutab *t, const char *path);
int ugs_otutab_write_biom(const ugs_otutab *t, const char *path);     /* -biomout: OTUTable::ToJsonFile json.cpp:32-103 */
int ugs_otutab_totals(const ugs_otutab *t, uint64_t *assigned, uint64_t *total);

/*
 * closed_ref sink (cmd_closed_ref searchcmd.cpp:11-19: usearch_global with -id 0.97 -stepwords 0, terminator 4/16;
 * ClosedRefSink::OnQueryDone closedrefsink.cpp:33-118).  _add takes one query's hits in HitMgr order with the label of
 * every hit's target and returns the -tabbedout line.  -dbotus / -dataotus are not built (the reference crashes on them).
 */
typedef struct ugs_closedref ugs_closedref;
ugs_closedref *ugs_closedref_create(void);
void ugs_closedref_destroy(ugs_closedref *c);
int ugs_closedref_add(ugs_closedref *c, const char *qlabel, const ugs_hit *hits, uint32_t n, const char *const *tlabels, char *line, int cap);
int ugs_closedref_totals(const ugs_closedref *c, uint64_t *assigned, uint64_t *unassigned, uint32_t *otus);


/* ------------------------------------------------------------------------------------------
 * cluster_fast (BASELINE config C3): the UCLUST greedy loop of ClusterFast (clusterfast.cpp:81-133).
 *
 * ugs_db_append = UDBData::AddSIToDB_CopyData (udbbuild.cpp:286-291; AddSeqNoncoded :256-284, AddWord/GrowRow :74-128) for n
 * sequences at once: they become targets nseq .. nseq+n-1, their letters are stored as given and the index rows of their
 * distinct valid words grow at the end - on the device.  Needs a database created with dbmask = 2 (an index over masked
 * letters cannot grow: the cluster database is never masked, SeqDB::FromFastx keeps the letters as read).  The small -> Big
 * ranking latch (udbusortedsearcher.cpp:39-58) follows the new size.  Batches uploaded before the call must be uploaded again.
 */
int ugs_db_append(ugs_db *db, const char *seqs, const uint64_t *offs, uint32_t n);

/* cmd_cluster_fast's searcher settings on top of ugs_params_init: terminator 1 accept / 8 rejects (terminator.cpp:10-14),
 * letters used as read (dbmask 2).  -id is required (makeclustersearcher.cpp:30-31). */
int ugs_params_set_cluster(ugs_params *p);

/*
 * The whole command on one GPU: DerepFull in input order (= the reference at -threads 1, derepfull.cpp:130-212,
 * derepresult.cpp:403-480; case-insensitive, both orientations with strand_both), then the greedy loop over the uniques in
 * input order (-sort unset).  The loop runs in batches against the centroid index as it stood when the batch started; the
 * centroids founded by earlier queries of the same batch are merged into every query's candidate walk exactly (word counts and
 * pair alignments from the device, the reference's ordering / cut-off / terminator rules replayed in input order), so the
 * result equals the serial loop's.  Results:
 *   seq_unique[nseq]      unique (derep cluster) of every input sequence; uniques are numbered by their first member
 *   uniq_seed[n_unique]   input index of a unique's first member (its letters and label stand for the unique)
 *   uniq_cluster[n_unique] cluster of every unique;   uniq_nhits[n_unique] 0 = founded its cluster, else its hits (1, or 2
 *                         with strand_both: HitMgr keeps one accept per strand)
 *   centroid_uniq[n_clusters], cluster_size[n_clusters] (input sequences incl. duplicates, ClusterSink::GetSize clustersink.cpp:123-150)
 *   hits[n_hits]          grouped by unique in order, each group in HitMgr::Sort order; .query = unique, .target = cluster
 */
typedef struct ugs_cluster ugs_cluster;
typedef struct ugs_cluster_stats {
  uint32_t batches;          /* device batches run                                                      */
  uint32_t batches_cut;      /* batches ended early because a query could not be replayed from the device's data */
  uint32_t max_batch;
  uint32_t units_heavy;      /* units ranked by the heavy-unit kernel (k_rank2<HV>: reads of abundant species, ugs_rank2.hip) - r6, was reserved_ */
  uint64_t queries_redone;   /* queries searched again in a later batch (cut batches, the small -> Big latch) */
  uint64_t inbatch_entries;  /* (query strand, earlier query of the batch) pairs with shared words that could matter */
  uint64_t pairs_in_batch;   /* of those, aligned on the device                                          */
  uint64_t hits_in_batch;    /* accepted hits whose target was founded inside the query's own batch       */
  uint64_t pairs_frozen;     /* pair alignments of the frozen-index walks                                 */
  uint64_t postings;         /* algorithmic postings of the frozen-index scans            */
  float    ms_rank, ms_align; /* summed device time of the frozen-index stages                            */
  /* host wall-clock seconds by stage: dereplication, frozen search (upload .. sync), in-batch counts (+ batch index),
   * device -> host copies, the two input-order passes, the pair stage, growing the index */
  float    s_derep, s_search, s_inbatch, s_d2h, s_replay, s_pairs, s_append, s_total;
} ugs_cluster_stats;
int ugs_cluster_fast(const ugs_params *p, const char *seqs, const uint64_t *offs, uint32_t nseq, int device, ugs_cluster **out);
/* The same with -sort length | size (GetSeqOrder clusterfast.cpp:37-79: the uniques are searched by decreasing seed length or
 * summed size= annotation - default 1, read whether or not -sizein is set, derepresult.cpp:211-225 - in QuickSortOrderDesc's
 * order) and -sizein (cluster sizes sum the annotations, ClusterSink::GetSize clustersink.cpp:119-150; a label without one fails
 * with UGS_E_ARG like the reference's "Missing size= in").  size_in[nseq] = ugs_label_size of every input label, may be NULL
 * when neither is used.  The uniques are then numbered in processing order (uniq_seed, uniq_*, hits.query follow it). */
#define UGS_SORT_NONE   0
#define UGS_SORT_LENGTH 1
#define UGS_SORT_SIZE   2
int ugs_cluster_fast_sorted(const ugs_params *p, const char *seqs, const uint64_t *offs, uint32_t nseq, int sort_mode,
                            const uint32_t *size_in, int sizein, int device, ugs_cluster **out);
/* GetSizeFromLabel label.cpp:152-161: (unsigned) atoi after the first ";size=", 0xffffffff = no annotation */
uint32_t ugs_label_size(const char *label);
void ugs_cluster_destroy(ugs_cluster *c);
int ugs_cluster_counts(const ugs_cluster *c, uint32_t *n_unique, uint32_t *n_clusters, uint64_t *n_hits, uint64_t *cigar_runs);
int ugs_cluster_get(const ugs_cluster *c, uint32_t *seq_unique, uint32_t *uniq_seed, uint32_t *uniq_cluster, uint32_t *uniq_nhits,
                    uint32_t *centroid_uniq, uint32_t *cluster_size, ugs_hit *hits, uint32_t *cigar_pool);
int ugs_cluster_get_stats(const ugs_cluster *c, ugs_cluster_stats *st);
/* -uc (S / H records per unique followed by its duplicates' records, then the C records: outputuc.cpp:10-93,
 * clustersink.cpp:477-493) and -centroids (by decreasing size in QuickSortOrderDesc's order, 80 columns: clustersink.cpp:262-289).
 * labels: the nseq input labels, NUL-terminated, concatenated in input order. */
int ugs_cluster_write_uc(const ugs_cluster *c, const char *labels, const char *path);
int ugs_cluster_write_centroids(const ugs_cluster *c, const char *labels, const char *path);
/* + MakeCentroidLabel clustersink.cpp:219-243: UGS_SIZEIN | UGS_SIZEOUT strip the size= annotation, UGS_SIZEOUT appends
 * ";size=<cluster size>;"; -minsize ends the file at the first smaller cluster (clustersink.cpp:275-277) */
#define UGS_SIZEIN  1
#define UGS_SIZEOUT 2
int ugs_cluster_write_centroids_sized(const ugs_cluster *c, const char *labels, const char *path, int size_flags, uint32_t minsize);

/* Page-lock / unlock a caller-owned host buffer (hipHostRegister): result buffers that are reused from batch to batch
 * are then filled by direct DMA instead of through the runtime's staging copies.  Optional; any host pointer works
 * with the fetch calls. */
int ugs_host_register(void *ptr, uint64_t bytes);
int ugs_host_unregister(void *ptr);

const char *ugs_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* UGS_H */
