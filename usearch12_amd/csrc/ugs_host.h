// ugs_host.h - handle structs shared by the host translation units (ugs_host.cpp, ugs_cluster.cpp).  Internal.
#pragma once
#include "ugs_dev.h"
#include "ugs_rank2.h"
#include <vector>
#include <mutex>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ugs_set_error("%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return UGS_E_HIP; } } while (0)
#define RCCHK(x) do { int rc_ = (x); if (rc_ != UGS_OK) return rc_; } while (0)

static const size_t LDS_MAX = 160 * 1024;

// Caller-owned host memory -> device WITHOUT handing a pageable caller pointer to the runtime (ugs_alloc.cpp).  hipMemcpy from pageable
// memory makes the runtime page-lock the caller's pages on the fly and lets the GPU read them in place; the one reproduction of round 5's
// silent abort (round 6, the round-5 tree on torch's bundled ROCm 7.0 runtime, 1 of 6 full suites) was a GPU memory fault at an address
// inside the HOST heap while the main thread was in ugs_db_create - i.e. in such a copy.  The library's own uploads of caller memory
// (database letters and offsets, tables, pair keys, ugs_db_append) therefore go through a page-locked staging buffer the library owns:
// memcpy on the host, DMA from the staging buffer, two slots in flight.  A source that IS page-locked (ugs_host_register, hipHostMalloc)
// is copied directly.  Ordered on `st`; the source may be reused when the call returns.
hipError_t ugs_h2d(void *dst, const void *src, size_t bytes, hipStream_t st);

// Debug / tuning switches, read from the environment ONCE per database handle, at ugs_db_create (never inside a search call).  Not part of the ABI: they exist
// for A/B measurements and fault isolation; the test-suite runs with none of them set unless a test names one.
struct UgsTune {
  int no_packed;          // UGS_NO_PACKED       k_align fetches every target from the byte array
  int longrows;           // UGS_LONGROWS        -1 unset, 0/1 force the LONG ranking instantiations off/on
  int gsize, gshift;      // UGS_GSIZE/UGS_GSHIFT partition size of k_rank (0 unset)
  int rank_wgs, align_wgs;// UGS_RANK_WGS_PER_CU / UGS_ALIGN_WGS_PER_CU  cap on resident workgroups (0 unset)
  long emit_limit;        // UGS_EMIT_LIMIT      candidate buffer of k_rank in keys (forces the regrow path in tests; 0 unset)
  int debug_sync;         // UGS_DEBUG_SYNC      finish every stage before the next, log
  int phase_clocks;       // UGS_PHASE_CLOCKS    print the kernels' phase clocks with the stats
  int wide_offsets;       // UGS_WIDE_OFFSETS    force the 64-bit-offset instantiations of the Big-path 4-bit ranking kernels
  int rank2;              // UGS_RANK2           -1 unset (= on where eligible), 0 off, 1 on
  bool qpk;                     // UGS_QPK=1                  nt query letters packed once by k_rank_setup for k_align (off: k_align packs them per unit; measured slower on)
  int align_group;              // UGS_ALIGN_GROUP            -1 unset (= 1), 0 off, n: rejects of a unit after which k_align tests its candidates four at a time
  int r2_g, r2_kcap, r2_waves; // UGS_R2_G / UGS_R2_KCAP / UGS_R2_WAVES  partition size, kept-key capacity, waves per CU of the bitmap kernel (0 unset)
  int r2_clcap;                 // UGS_R2_CLCAP               chunk descriptors per window of the bitmap kernel (0 unset)
  int setup_stream;             // UGS_SETUP_STREAM=1         k_rank_setup of a search on a second stream (overlaps the kernels of the batch in front)
  int r2_hv;                    // UGS_R2_HV                  -1 unset (= on), 0: cluster_fast's deferred units go straight to k_rank (A/B), 2: every unit through the heavy-unit kernel (tests)
  int r2_p16;                   // UGS_R2_P16                 -1 unset (= on), 0: the bitmap kernel streams the 32-bit postings (A/B)
  int r3, r3_sp, r3_pps;        // UGS_R3 / UGS_R3_SP / UGS_R3_PPS  sparse index: -1 unset (= k_rank3g), 0 = k_rank2g; k_rank3g: partitions per super-partition
                                //                            (0 unset = per unit, from its postings), postings per super-partition aimed at (0 unset = 4096)
};
UgsTune ugs_tune_read();

struct ugs_db {
  ugs_params p;
  UgsTune tune;                     // the environment's debug switches as they stood at ugs_db_create
  int device;
  hipStream_t stream;
  hipStream_t setup_stream;         // UGS_SETUP_STREAM=1: the unit set-up kernels of a search run here, beside the kernels of the batch in front (null otherwise)
  int num_cu;
  UgsDbView v;
  // owned device memory
  uint8_t *d_seqs; uint64_t *d_offs; uint64_t *d_row_off; uint32_t *d_postings; uint32_t *d_part;
  uint32_t *d_part2; uint64_t part2_cap;   // dense Big-path indexes: the partition table of the bitmap ranking kernel (ugs_rank2.hip)
  uint16_t *d_post16; uint64_t post16_cap; // ... and the postings as 16-bit offsets inside their partition (built on first use by a plain search: plan_launch)
  std::mutex post16_mu;                    // (the copy is made inside a search plan: two host threads may plan batches of one handle)
  uint64_t index_gen, post16_gen;          // the index as it stands (counts ugs_db_replan calls) / the one d_post16 was made from
  uint2 *d_pk; uint64_t pack_cap;   // nt: 2-bit letters + "other" bits, one uint2 per 16 letters (ugs_dev.h UgsDbView::pk)
  uint32_t *d_step; UgsTables *d_tab;
  std::vector<uint32_t> step;       // host copy: step[Nu]
  uint64_t n_postings, hbm_bytes;
  uint32_t max_row, max_tlen;
  // usearch_local
  int8_t *d_xsub2; uint8_t *d_xcls;
  UgsLocalView lv;
  // pair filters / -abskew
  uint32_t *d_tkey, *d_tsize; bool have_tkey, have_tsize;
  bool sparse;                      // sparse dictionary (protein): short index rows
  bool r2_gather;                   // part2 was built for the gather variant of the bitmap kernel (k_rank2g: sparse Big-path index)
  // capacities (elements) of the growable arrays and the letter count: ugs_db_append grows the DB in place
  uint64_t nletters, seq_cap, off_cap, post_cap, part_cap;
  uint32_t gsize_limit;             // small ranking path only: cap on the partition size (0 = none), see ugs_cluster.cpp
  uint64_t *d_row_off2; uint32_t *d_postings2; uint64_t post_cap2;   // spare index arrays (ugs_db_append merges into them, then swaps)
};

struct ugs_batch {
  ugs_db *db;
  uint32_t max_queries; uint64_t max_letters;
  uint32_t nq, max_qlen, K, nstrand;
  UgsBatchView v;
  uint8_t *d_qseqs; uint64_t *d_qoffs;
  uint32_t *d_qn, *d_qoff; ugs_hit *d_compact; void *d_scan_tmp; size_t scan_tmp_bytes;
  uint32_t *d_cand, *d_cand_cnt, *d_cand_n, *d_hit_n, *d_cigar, *d_runs;
  ugs_hit *d_hits; uint64_t *d_emit; uint8_t *d_tb;
  uint32_t *d_unit_ns, *d_unit_slots; uint64_t unit_slots_alloc;
  void *d_qpk; uint64_t qpk_alloc; uint32_t qpk_stride;          // packed query planes (nt), see UgsBatchView::qpk
  uint32_t *d_defer;                // units the bitmap ranking kernel hands on to k_rank
  uint32_t *d_defer2;               // cluster_fast: ... and the ones the heavy-unit kernel behind it hands on (allocated on first use)
  UgsRank2Params r2; int r2_grid;   // its launch (r2_grid == 0: not used for this batch)
  // usearch_local
  uint32_t hit_slots;               // hit table entries per unit
  int2 *d_qthr; uint8_t *d_ltb; uint2 *d_lrow; uint32_t *d_lruns;
  uint64_t ltb_alloc, lrow_alloc, lruns_alloc;
  UgsLocalView lv; int lgrid, lwpb; size_t llds;
  uint32_t *d_qkey, *d_qsize; bool have_qkey, have_qsize;
  unsigned long long *d_cigar_used, *d_ctr;
  uint64_t cigar_cap, emit_cap_alloc, tb_alloc, runs_alloc;
  uint64_t emit_limit;              // keys per workgroup the candidate buffer is sized for (grown on demand by ugs_batch_sync)
  int rank_grid_alloc, align_waves_alloc;
  UgsRankLaunch rl; UgsAlignLaunch al;
  hipEvent_t ev0m;                  // the ranking kernels begin (= ev0s unless the set-up runs on a stream of its own)
  hipEvent_t ev0, ev0s, ev0r, ev1, ev2;   // ev0r: end of the bitmap ranking kernel (before k_rank takes the deferred units)
  bool r2_ran;
  bool cl_mode;                     // the batch of a cluster_fast loop (ugs_cluster.cpp): its searches leave walk records, the bitmap kernel runs its CL instantiation
  // upload path: H2D copies go through the batch's own copy stream; the search waits for ev_up on the handle's stream,
  // so the upload of one batch overlaps the kernels of another (h_rel: page-locked staging of the relative offsets)
  hipStream_t copy_stream; hipEvent_t ev_up, ev_done; uint64_t *h_rel;   // ev_done: end of the last enqueued search
  uint32_t compact_base;            // query base of the grouped hit table in d_compact (query_base after a search)
  uint32_t query_base;              // ugs_batch_set_query_base: what the search's own grouping adds to ugs_hit.query (a shard's offset)
  bool searched, synced;
  // deep walks (UGS_A_DEEP, ugs_deep.hip): parked walks, the scratch of their complete candidate lists, overflow hit blocks
  UgsWalkState *d_walk_state; uint32_t *d_open_list;
  uint32_t *d_deepU, *d_deepR; uint64_t deep_scr_alloc; int deep_grid;
  bool deep_dirty;                  // d_deepU may hold counts of a pass that did not reach its end (zeroed before the next one)
  uint32_t *d_keyn; uint64_t *d_koff; uint64_t keyn_alloc;
  uint64_t *d_keys, *d_keys_sorted; uint64_t keys_alloc; void *d_sort_tmp; size_t sort_tmp_bytes;
  ugs_hit *d_xpool; uint32_t *d_xnext; uint32_t xblocks_cap; unsigned long long *d_xblocks_used;
  uint64_t compact_alloc;           // entries of d_compact
  uint64_t deep_units, deep_keys_total;   // diagnostics of the last search: parked units / keys of their lists
  unsigned long long ctr[UGS_CTR_N];
  unsigned long long cigar_used_host;
  uint64_t q_letters;
};



// internal entry points of ugs_host.cpp used by the cluster_fast driver
int ugs_db_replan(ugs_db *db);                    // partition size / table + Big latch after the index changed
void ugs_qs_order_desc(const float *V, int left, int right, unsigned *Order);
