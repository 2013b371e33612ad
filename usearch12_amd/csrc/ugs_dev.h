// ugs_dev.h - internal device-side views shared by the HIP translation units.
// Product code for gfx950 only (wave64, 160 KiB LDS/CU); see DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ugs.h"

// UGS_KERNEL_LOG=1 (debug, read when the library is loaded): every kernel launch of the library - rocPRIM's included - is named on stderr,
// waited for, and named again when it has finished: the last "launched" line without its "done" is the kernel a GPU fault belongs to
// (with UGS_GUARD_ALLOC=1 and UGS_ABORT_BT: which buffer, allocated where, overrun by which kernel).  ugs_alloc.cpp.
extern int ugs_kernel_log;
void ugs_after_launch(const char *kernel, hipStream_t st);
void ugs_before_launch(const char *kernel);
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                                          \
  do {                                                                                                                              \
    if (ugs_kernel_log) ugs_before_launch(#kernelName);                                                                             \
    hipLaunchKernelGGLInternal((kernelName), (numBlocks), (numThreads), (memPerBlock), (streamId), __VA_ARGS__);                    \
    if (ugs_kernel_log) ugs_after_launch(#kernelName, (streamId));                                                                  \
  } while (0)

#define UGS_WAVE 64
#define UGS_MAXREPS 8          // hspfinder.h:10
#define UGS_XLUT_D 17          // k_align: x-drop deficits (in units of 2 half-units) the extension table covers: X <= 32 half-units
#define UGS_ALIGN_HDR (2112 + UGS_XLUT_D * 16 * 2)     // workgroup-shared LDS in front of k_align's per-wave regions: letter tables + the extension table
#define UGS_BAD_WORD 0xffffffffu
#define UGS_KMAX 64            // max candidates walked per strand (max_accepts+max_rejects-1)

// Per-DB constant tables (built on the host from the alphabet rules).
struct UgsTables {
  uint8_t udb_letter[256];   // UDB word letter, 0xff = voids the word (invalid or lower-case) udbparams.cpp:540-555
  uint8_t hsp_letter[256];   // HSP-finder letter, invalid -> 0                               hspfinder.cpp:238-248
  uint8_t cls[256];          // score/identity class: letter 0..25 | 32 if lower-case; 31 = non-alpha
  uint8_t comp[256];         // reverse-complement byte map (seqinfo.cpp:292-323)
  int8_t  sub2[32 * 32];     // 2 x substitution score by letter (setnucmx.cpp / blosum62.cpp)
  uint64_t match[64];        // identity bitmask row per class (alpha2.cpp:220-300)
};

struct UgsDbView {
  const uint8_t  *seqs;      // masked DB letters
  const uint2 *pk;           // nt only (null otherwise): the same letters packed, one uint2 per 16 letters by GLOBAL letter index (letter i of
                             // the array -> bits 2(i%16).. of entry i/16): .x = 2 bits per letter (A,C,G,T/U = 0..3, anything else 0),
                             // .y = the "anything else" bits in the same layout; the two words side by side so that a target is one stream
  const uint64_t *offs;      // [nseq+1]
  uint32_t nseq;
  uint32_t slots;
  const uint64_t *row_off;   // [slots+1]
  const uint32_t *postings;  // target indexes, ascending inside a row
  const uint32_t *part;      // [slots*(np+1)] offset (relative to row start) of first posting with target >= p*gsize
  uint32_t np;               // number of target partitions
  uint32_t gsize;            // targets per partition (multiple of 64)
  const uint32_t *part2;     // dense Big-path indexes only (null otherwise): the same table for the bitmap kernel's larger partitions (ugs_rank2.hip)
  uint32_t np2, gsize2;      // its partition count and size (a multiple of 8192, <= 65536)
  const uint32_t *step_tab;  // [step_n] Big-path QueryStep for Nu unique words (wordparams.cpp:167-192)
  uint32_t step_n;
  const UgsTables *tab;
  int32_t word_len;          // UDB word length
  int32_t alpha;             // 4 / 20
  int32_t big;               // Big ranking path (nseq > -big)
  uint32_t bump_pct;
  // aligner constants in half-score units
  int32_t hsp_w;             // HSP finder word length
  int32_t hsp_words;         // alpha^hsp_w
  int32_t xdrop2;            // 2 * xdrop_nw
  int32_t minscore2;         // smallest half-unit score >= MinGlobalHSPScore
  float   min_hsp_fract_id;  // MinGlobalHSPFractId (float compare, getglobalhsps.cpp:57)
  int32_t min_hsp_len_opt;   // -minhsp
  int32_t band;
  int32_t open2, ext2, topen2, text2;   // internal / terminal gap penalties x2 (alnparams.cpp:380-384)
  double  id_accept;
  int32_t id_set;
  // optional accept filters (Accepter::IsAcceptLo accepter.cpp:41-91), UGS_F_* bits
  uint32_t filter_mask;
  float maxid, query_cov, max_query_cov, target_cov, max_target_cov;
  uint32_t mincols, maxgaps, maxdiffs, mindiffs;
  int32_t max_accepts, max_rejects;   // max_accepts: hit slots per unit AND (outside deep walks) the walk's accept limit; max_rejects: the reject limit
  int32_t acc_limit;         // the walk's accept limit where it differs from the slots (deep walks: UGS_A_DEEP): 0x7fffffff = unlimited
  int32_t is_nucleo;
  uint32_t max_tlen;
  // pair filters (Accepter::RejectPair accepter.cpp:140-197) and -abskew: UGS_P_* bits, values, per-target keys
  uint32_t pair_mask;
  float min_sizeratio, minqt, maxqt, minsl, maxsl, abskew;
  const uint32_t *t_key, *t_size;
  uint32_t align_flags;      // UGS_A_FULLDP | UGS_A_GAFORCE | UGS_A_TERMID | UGS_A_TERMIDD
  float termid, termidd;
  uint32_t group_after;      // k_align: rejects of a unit after which its candidates go through the group filter (0 = never)
};

struct UgsBatchView {
  const uint8_t  *qseqs;
  const uint64_t *qoffs;     // [nq+1]
  uint32_t nq;
  uint32_t nstrand;          // 1 or 2
  uint32_t K;                // candidates kept per unit
  uint32_t max_qlen;
  // ranking outputs, per unit (= query*nstrand + strand)
  uint32_t *cand;            // [units*K]
  uint32_t *cand_cnt;        // [units*K]
  uint32_t *cand_n;          // [units]
  // sampled index rows per unit, written by k_rank_setup and read by k_rank
  uint32_t *unit_ns;         // [units]
  uint32_t *unit_slots;      // [units * ns_max]
  // nt: the unit's letters (strand applied) packed 2 bits each + the "other letter" plane, written once by k_rank_setup, read by k_align
  // (BASELINE north_star "query batches packed 2-bit"): word k of unit u at qpk[u * qpk_stride + k] = {letters, other-letter bits}; null = k_align packs
  uint2 *qpk; uint32_t qpk_stride;
  // ranking scratch: per resident workgroup
  uint64_t *emit_buf;        // [rank_wgs * emit_cap]
  uint64_t emit_cap;
  // alignment outputs
  ugs_hit  *hits;            // [units*max_accepts]
  uint32_t *hit_n;           // [units]
  uint32_t *cigar_pool;      // run pool
  uint64_t cigar_cap;
  unsigned long long *cigar_used;   // device counter (demand, may exceed cap => overflow)
  // alignment scratch: per resident wave
  uint8_t  *tb;              // [align_waves * tb_stride]
  uint64_t tb_stride;
  uint32_t *runs;            // [align_waves * runs_stride]
  uint32_t runs_stride;
  // counters: [0]=postings [1]=target letters [2]=pairs [3]=dp cells [4]=hits [5]=error flags
  unsigned long long *counters;
  const uint32_t *q_key, *q_size;   // per query: label key / ;size= annotation (pair filters, -abskew)
  // cluster_fast (ugs_cluster.cpp); all null outside it
  uint64_t *cand_key;        // [units*K] full ranking key (count, first-touch position) of every candidate
  uint64_t *cl_ev;           // [units*UGS_CL_EV] strict prefix maxima of the scan: count << 44 | position, descending
  uint32_t *cl_info;         // [units*4] M, NextValue, number of prefix maxima, -
  uint32_t *walk_n;          // [units] candidates the alignment walk visited
  const uint32_t *unit_map;  // [units] pair stage: unit -> query << 1 | strand (several units per query)
  // cluster_fast: the ranking kernels take the units in descending order of their postings (a batch of reads holds a few units a hundred
  // times heavier than the rest: started last they are the tail of the launch).  k_unit_cost writes unit_cost and a histogram of cost
  // classes, k_unit_order the order; all null outside cluster_fast
  uint32_t *unit_cost;       // [units] sampled postings of the unit (saturated)
  uint32_t *unit_order;      // [units] units by descending cost class
  uint32_t *order_hist;      // [512] units per cost class | cursors (zeroed by the launcher)
  uint32_t *defer_list;      // [units] units the bitmap ranking kernel (ugs_rank2.hip) hands on to k_rank; counters[UGS_CTR_DEFER] of them
  uint32_t use_defer;        // k_rank (HOT instantiation): take the units from defer_list instead of 0 .. units-1 (1: counters[UGS_CTR_DEFER] of
                             // them; 2: counters[UGS_CTR_DEFER2] - the list the heavy-unit kernel leaves behind, cluster_fast)
  // deep walks (UGS_A_DEEP: max_accepts + max_rejects - 1 > UGS_KMAX, or one of them unlimited; ugs_deep.hip).  First pass: a walk that
  // used up a FULL list of K candidates without meeting a limit is parked - its counters in walk_state, its unit in open_list
  // (counters[UGS_CTR_OPEN] of them).  Continuation pass (walk_units != null): k_align takes unit walk_units[i], restores the counters
  // and walks on through the unit's COMPLETE sorted candidate list deep_keys[deep_off[i] .. deep_off[i + 1]) from the first one it
  // has not visited, a page of 64 at a time.  Accepted hits beyond a unit's slots go to blocks of UGS_XBLOCK hits chained per unit.
  struct UgsWalkState *walk_state;  // [units]
  uint32_t *open_list;              // [units]
  const uint32_t *walk_units; uint32_t n_walk;
  const uint64_t *deep_keys; const uint64_t *deep_off;
  ugs_hit *xpool; uint32_t *xnext; uint32_t xblocks_cap; unsigned long long *xblocks_used;   // overflow hit blocks: pool, chain links, capacity, demand
};
struct UgsWalkState { uint32_t nacc, nrej, nvis, xhead, xcur, pad0, pad1, pad2; };   // xhead / xcur: first / current overflow block (0xffffffff none)
#define UGS_XBLOCK 64u
#define UGS_CL_EV 16
#define UGS_A_NOTERM 0x100u  // internal align flag: rejects never end a walk (the in-batch pair stage of cluster_fast)
#define UGS_A_OPENWALK 0x200u // internal: maxaccepts or maxrejects is 0 (unlimited)
#define UGS_A_DEEP 0x400u     // internal: walks may need more than the UGS_KMAX candidates a ranking pass keeps (see UgsBatchView::walk_state)

enum { UGS_CTR_POSTINGS = 0, UGS_CTR_TLETTERS, UGS_CTR_PAIRS, UGS_CTR_CELLS, UGS_CTR_HITS, UGS_CTR_ERR,
       UGS_CTR_T0, UGS_CTR_T1, UGS_CTR_T2, UGS_CTR_T3, UGS_CTR_T4, UGS_CTR_T5, UGS_CTR_T6, UGS_CTR_T7, UGS_CTR_NEXT_UNIT, UGS_CTR_NEXT_RANK, UGS_CTR_NEXT_SETUP, UGS_CTR_EMIT_MAX, UGS_CTR_NEXT_RANK2, UGS_CTR_DEFER, UGS_CTR_R2_DONE, UGS_CTR_GROUPED, UGS_CTR_OPEN,
       UGS_CTR_NEXT_HEAVY, UGS_CTR_DEFER2, UGS_CTR_HV_DONE, UGS_CTR_N };  // T*: phase clocks (profiling); EMIT_MAX: most keys one wave emitted for one unit (set when UGS_ERR_EMIT is)
enum { UGS_ERR_NS = 1, UGS_ERR_HSPCAP = 2, UGS_ERR_RUNS = 4, UGS_ERR_EMIT = 8, UGS_ERR_LOCAL = 16, UGS_ERR_LOCAL_HITS = 32, UGS_ERR_PAIRCAP = 64, UGS_ERR_XHITS = 128 };   // XHITS: the overflow hit blocks of a deep walk ran out (the host grows the pool and runs the pass again)

// usearch_local (ugs_local.hip): x-drop tables and scratch, per-query score gates
struct UgsLocalView {
  const int8_t *sub2; const uint8_t *cls;   // x-drop score table / letter class (ugs_xdrop_tables)
  int open2, ext2; float xdrop_g, abs_open, abs_ext, xdrop_u;
  uint32_t seed_w;                          // seed word length (-hspw)
  const int2 *qthr;                         // [nq] smallest ungapped / gapped score (half-units) that passes the e-value gates
  uint32_t W;                               // LDS row width of the x-drop rows
  uint8_t *tb; unsigned long long tb_cap;   // per wave
  uint2 *rowinfo; uint32_t rows_cap;        // per wave
  uint32_t *runbuf; uint32_t runbuf_cap;    // per wave: forward runs, backward runs, merged path
  uint32_t hit_slots;                       // hit table entries per unit
  uint32_t seed_cap;                        // seeds listed per round (>= max_qlen + 64)
};

// launch descriptors computed on the host
struct UgsRankLaunch { int bits; int wpb; int grid; size_t lds; uint32_t ns_max; uint32_t part_words; int fast8; int longrows; int wide; int debug_sync; };
struct UgsAlignLaunch { int wpb; int grid; size_t lds; uint32_t hsp_cap; uint32_t seed_cap; };

// kernels' host-callable launchers (defined in the .hip files)
int ugs_launch_mask(uint8_t *d_seqs, const uint64_t *d_offs, uint32_t nseq, int dbmask, hipStream_t st);
int ugs_launch_pack(const UgsTables *d_tab, const uint8_t *d_seqs, uint64_t word_lo, uint64_t word_hi, uint2 *d_pk, hipStream_t st);
int ugs_build_index(const UgsTables *d_tab, const uint8_t *d_seqs, const uint64_t *d_offs, uint32_t nseq,
                    uint64_t nletters, int word_len, int alpha, uint32_t slots, uint64_t **d_row_off,
                    uint32_t **d_postings, uint64_t *n_postings, uint32_t *max_row, hipStream_t st);
int ugs_build_part(const uint64_t *d_row_off, const uint32_t *d_postings, uint32_t slots, uint32_t np,
                   uint32_t gsize, uint32_t *d_part, hipStream_t st);
int ugs_rank_blocks_per_cu(int threads, size_t lds, int big, int bits, int fast8, int longrows, int wide);
// The ranking kernels' instantiations: ONE table for rank_kernel()'s ordinals, the "compiled" mask of ugs_rank_instances_seen, the names
// ugs_debug_rank_instance_name hands to the test-suite (tests/test_zz_gpu_coverage.py holds no list of its own).
#define UGS_RANK_INST_TABLE(X) \
  X(0, BIG4, "Big 4-bit (HOT)") X(1, BIG4_LONG, "Big 4-bit long rows") X(2, BIG_FLAT, "Big 8/16-bit flattened (sparse)") \
  X(3, BIG_DENSE, "Big 8/16-bit dense") X(4, BIG_DENSE_LONG, "Big 8/16-bit dense, long rows") \
  X(5, SMALL4, "small 4-bit") X(6, SMALL4_LONG, "small 4-bit long rows") X(7, SMALL_FLAT, "small 8/16-bit flattened") \
  X(8, SMALL_DENSE, "small 8/16-bit dense") X(9, SMALL_DENSE_LONG, "small 8/16-bit dense, long rows") \
  X(12, BIG4_WIDE, "HOT, 64-bit offsets") X(13, BIG4_LONG_WIDE, "long rows, 64-bit offsets") \
  X(14, R2, "k_rank2 (bitmap)") X(15, R2G, "k_rank2g (bitmap, sparse index)") X(16, R2_CL, "k_rank2, cluster_fast instantiation") \
  X(17, R3G, "k_rank3g (two filter passes, sparse index)") X(18, R2_P16, "k_rank2 over 16-bit postings") \
  X(19, R2_HV, "k_rank2, heavy units of cluster_fast (4-bit counters, two passes)")
#define UGS_RANK_INST_ENUM(i, n, s) UGS_RI_##n = i,
enum { UGS_RANK_INST_TABLE(UGS_RANK_INST_ENUM) UGS_RI_END };
unsigned long long ugs_rank_instances_seen(unsigned long long *compiled);
const char *ugs_rank_instance_name(int ordinal);          // null: no such instantiation
int ugs_rank_is_hot(int big, int bits, int fast8, int longrows);
int ugs_align_blocks_per_cu(int threads, size_t lds, int is_nucleo);
size_t ugs_rank_fixed_lds(uint32_t ns_max, uint32_t max_qlen, uint32_t part_words, int hot);
int ugs_compact_hits(const uint32_t *d_hit_n, const ugs_hit *d_table, uint32_t nq, uint32_t ns, uint32_t ma,
                     uint32_t *d_qn, uint32_t *d_qoff, ugs_hit *d_out, void *d_tmp, size_t tmp_bytes, uint32_t query_base,
                     hipStream_t st);
// the same in two steps for tables with overflow blocks (deep walks): counts + offsets first (the caller sizes d_out by their total), then the copy
struct UgsXHits { const UgsWalkState *state; const ugs_hit *pool; const uint32_t *next; };
int ugs_count_hits(const uint32_t *d_hit_n, uint32_t nq, uint32_t ns, uint32_t *d_qn, uint32_t *d_qoff, void *d_tmp, size_t tmp_bytes, hipStream_t st);
int ugs_copy_hits(const uint32_t *d_hit_n, const ugs_hit *d_table, uint32_t nq, uint32_t ns, uint32_t ma, const uint32_t *d_qoff, ugs_hit *d_out,
                  uint32_t query_base, const UgsXHits *x, hipStream_t st);
// deep walks (ugs_deep.hip): the complete sorted candidate lists of the parked units
struct UgsDeepArgs {
  const uint32_t *units; uint32_t n_units;       // the parked units of this chunk
  uint32_t *U, *R; uint64_t stride;                // per workgroup: touch counts (kept zero between units) and first-touch rows of all targets
  uint32_t *key_n;                               // [n_units] kept keys per unit (mode 0 writes, mode 1 uses as cursor base 0)
  const uint64_t *key_off; uint64_t *keys;       // mode 1: unit i's keys go to keys[key_off[i] ..)
  uint32_t ns_max; int mode;
};
int ugs_launch_deep(const UgsDbView &db, const UgsBatchView &b, const UgsDeepArgs &a, int grid, hipStream_t st);
int ugs_deep_sort(uint64_t *d_keys, uint64_t *d_sorted, uint64_t total, uint32_t segments, const uint64_t *d_off, void **d_tmp, size_t *tmp_bytes, hipStream_t st);
size_t ugs_compact_tmp_bytes(uint32_t nq);
struct UgsRank2Params;
// st_setup (optional): the stream the unit set-up kernels run on - when it differs from st, the ranking kernels on st wait for
// ev_setup_done; ev_rank_start (optional) is recorded on st where the ranking kernels begin
int ugs_launch_rank(const UgsDbView &db, const UgsBatchView &b, const UgsRankLaunch &L, hipStream_t st, hipEvent_t ev_setup_done,
                    const UgsRank2Params *r2 = nullptr, int r2_grid = 0, hipEvent_t ev_r2_done = nullptr,
                    hipStream_t st_setup = nullptr, hipEvent_t ev_rank_start = nullptr);
int ugs_launch_align(const UgsDbView &db, const UgsBatchView &b, const UgsAlignLaunch &L, hipStream_t st);
size_t ugs_local_wave_lds(uint32_t W, uint32_t max_qlen, uint32_t max_tlen, uint32_t seed_cap);
int ugs_local_blocks_per_cu(int threads, size_t lds);
int ugs_launch_local(const UgsDbView &db, const UgsBatchView &b, const UgsLocalView &lv, int grid, int wpb, size_t lds, hipStream_t st);
void ugs_set_error(const char *fmt, ...);
// device memory (ugs_alloc.cpp): hipMalloc / hipFree, or - UGS_GUARD_ALLOC=1 - a mapping per buffer, right-aligned against an unmapped page
hipError_t ugs_malloc_at(void **p, size_t bytes, const char *file, int line);
hipError_t ugs_free(void *p);
template <class T> static inline hipError_t ugs_malloc_t(T **p, size_t bytes, const char *file, int line) { return ugs_malloc_at((void **)p, bytes, file, line); }
#define ugs_malloc(p, n) ugs_malloc_t((p), (n), __FILE__, __LINE__)      // (the call site goes into the guard allocator's table)
void ugs_xdrop_tables(int is_nucleo, float m2, float mm2, int8_t sub2[1024], uint8_t cls[256]);   // ugs_xdrop.hip
extern const char UGS_B62_ORDER[];          // the 23 alphabetic BLOSUM62 symbols
extern const signed char UGS_B62[23][23];
