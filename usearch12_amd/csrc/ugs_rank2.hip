// ugs_rank2.hip - the Big-path ranking kernel for dense indexes ("bitmap kernel"), gfx950.
//
// Replaces the same reference code as k_rank's Big path (ugs_rank.hip):
//   UDBUsortedSearcher::UDBSearchBig  scan + first-touch list            udbusortedsearcherbig.cpp:82-100
//   CountSortSubsetDesc (+ the NextValue / MinValue = prevMax/2 cut-off)   countsort.cpp:110-191
//   GetWordCountingParams' sampled words come from k_rank_setup            wordparams.cpp:167-192
//
// What the reference computes per query: U[t] = number of sampled index rows that hold target t, the targets in
// first-touch order (row-major over the sampled rows, targets ascending inside a row), a stable descending counting
// sort of them, of which the candidate loop can consume at most K.  So the result is the K smallest keys
// (count desc, first row asc, target asc) plus the cut-off that depends on the first position of every count value.
//
// Design (DESIGN.md section 3 "K-rank2"): ONE WAVE PER UNIT (query x strand), no workgroup barriers anywhere.
//   * the target space is cut into partitions of G targets (G <= 65536); the wave keeps ONE BIT per target of the
//     current partition in LDS (8 KB) instead of a 4-bit counter (k_rank: 5.6 KB for 11 264 targets), so a partition
//     is six times larger, a (row, partition) sub-row holds ~220 postings instead of ~39 and is read as ONE 16-byte load
//     per lane (four consecutive postings): 1 KB per load instruction, sub-rows cover their cache lines, and the
//     partition-table look-ups per unit drop from 979 to 187
//   * the rows of a partition are walked in DESCENDING row order with one ds_or_rtn per posting: a posting whose bit
//     was already set is a "second or later touch" and leaves a RECORD (row, target).  A target with count c leaves
//     c - 1 records, the last of them from its LOWEST row = its first-touch row: count and first touch of every
//     target with count >= 2 follow from the records alone (count-1 targets never matter individually: the fill walks
//     the first rows against the selected list, as k_rank's big_path_fill)
//   * posting loads run D chunks ahead of their use (ring of D register quads, the compiler's counted s_waitcnt vmcnt)
//   * when a partition is done its ~44 records are grouped by target (two 2048-bit hash filters tell the few
//     records that may share a target from the rest; those are compared all-pairs with readlane), turned into 32-bit
//     sortable keys [15 - count : 4][row : 4][target : 24] and PRUNED: a count-2 key is dropped when K count-2 keys of
//     earlier partitions (smaller targets) with the same or a lower row are already kept - it cannot be among the K
//     smallest, and the smallest key of every count value (all the cut-off needs) is never dropped
//   * the few kept keys (~100) are ranked all-pairs at the end of the unit
// A unit outside this kernel's envelope (more than 15 sampled rows, more than 128 records in one partition, more kept
// keys than the LDS list holds) is DEFERRED: its index goes to a list that the general kernel (k_rank, ugs_rank.hip)
// processes right behind this one.  Never a different result, never a CPU path.
#include "ugs_dev.h"
#include "ugs_rank2.h"
#include <cstdlib>
#include <cstdio>
#include <algorithm>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ugs_set_error("%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return UGS_E_HIP; } } while (0)

#include "ugs_ring_dev.h"

#define R2_SCAP 128u            // records of one partition the grouping handles (two register batches)
#define R2_SCAP_CL 256u         // ... of the cluster_fast instantiation (four: a read's species has many centroids)
#ifndef R2_HB_BITS
#define R2_HB_BITS 2048u        // bits of each of the two hash filters
#endif
#define R2_KEY_INF 0xffffffffu
// 32-bit words of k_rank2's LDS carve between the bitmap and the chunk list: staging records, the hash filter, s_c2 / s_cum / s_fpk /
// s_slots (16 each), s_rs (16 x u64), s_pe (32).  ONE definition for the kernel's carve and the host's size (ugs_rank2_lds).
__host__ __device__ constexpr uint32_t r2_fixed_words(bool cl) { return (cl ? R2_SCAP_CL : R2_SCAP) + R2_HB_BITS / 32u + 16u * 4u + 32u + 32u; }
#ifndef R2_V2
#define R2_V2 1                 // the ring's stage body, second version (r5); 0 = the r4 body (A/B builds)
#endif
#ifndef UGS_R2_DEPTH
#define UGS_R2_DEPTH 4          // ring slots: posting chunks in flight per wave (3 behind the one being counted)
#endif


// one chunk of a sub-row (<= 256 consecutive postings), 8 bytes in the unit's LDS chunk list:
//   lo = low 32 bits of the chunk's first element index in the postings array
//   hi = n (1..256 valid postings; 0 = padding) | row << 9 | partition << 13 (11 bits) | high 8 bits of the element index << 24
#define R2D_N(m) ((m) & 511u)
#define R2D_ROW(m) (((m) >> 9) & 15u)
#define R2D_PART(m) (((m) >> 13) & 2047u)
#define R2D_AHI(m) ((m) >> 24)
// P16 (16-bit partition-relative postings, r6): a chunk = <= 256 consecutive u16 ELEMENTS starting at a multiple of four elements (8-byte
// loads per lane); the first `lead` (0..3) of them belong to the sub-row in front and are masked:
//   lo = low 32 bits of the chunk's first element index (a multiple of 4)
//   hi = span (lead + valid postings, 1..256; 0 = padding) | row << 9 | partition << 13 (9 bits) | lead << 22 | high 8 bits of the index << 24
#define R2D_PART16(m) (((m) >> 13) & 511u)
#define R2D_LEAD16(m) (((m) >> 22) & 3u)
#define R2_P16_MAXNP 512u
// HV (the heavy-unit instantiation, r6): 10 bits of partition, bit 23 = the chunk belongs to the partition's SECOND pass
#define R2D_PART_HV(m) (((m) >> 13) & 1023u)
#define R2D_PASS2_HV(m) (((m) >> 23) & 1u)
#define R2_HV_MAXNP 1024u

// LDS byte address of target t's bitmap word: ((t >> 5) << 2) + nsub8 as a shift and ONE v_lshl_add_u32 (the compiler's own choice
// for the expression is shift, and, add)
__device__ __forceinline__ uint32_t r2_word_addr(uint32_t t, uint32_t nsub8)
{
  uint32_t a;
  asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a) : "v"(t >> 5), "v"(nsub8));
  return a;
}

// what the emission of a chunk needs after its atomics were issued
struct R2Stage { uint32_t old[4], bit[4], t[4]; };

// CL: the instantiation for cluster_fast (ugs_cluster.cpp).  Its host side merges a unit's walk with the centroids founded inside the same
// batch, so the candidates are chosen WITHOUT the MinValue cut-off and leave their full keys (cand_key), and the strict prefix maxima
// of the scan go out as cl_ev / cl_info - exactly what k_rank writes in that mode (ugs_rank.hip "cluster_fast").  A centroid index gives
// a read hundreds of targets with count >= 3 (the centroids of its species), so a full kept-key list is not a reason to defer there:
// the list is COMPACTED to its K smallest keys (the smallest key of every count value is tracked apart, s_fpk, as the keys are made).
#define R2_CMAXV 4095u          // ugs_rank.hip make_key: (CMAXV - count) << POS_BITS | first row << 32 | target
#define R2_POS_BITS 44
// P16: the scan streams UgsRank2Params::post16 - the index's postings as 16-bit offsets inside their partition (target mod G), same
// element positions as UgsDbView::postings - instead of the 32-bit targets: half the bytes of the dominant stream, four postings per
// 8-byte load and lane (the same lane occupancy and instruction count per posting as the 16-byte loads of four 32-bit targets).
// HV (r6): the instantiation for the HEAVY units of cluster_fast - the units the CL instantiation gave up because a partition held more
// second touches than its record list (a read of an abundant species: hundreds to thousands of centroids share most of its words).
// Records do not scale there (a target with count c leaves c - 1 of them); this instantiation COUNTS instead:
//   * partitions of UgsDbView::gsize targets (k_rank's table, ~ 9 000 targets), a 4-bit counter per target in LDS (counts <= ns <= 15);
//   * every partition is streamed twice through the same ring: pass 1 adds 1 to the counter of every posting (ds_add, nothing comes
//     back), pass 2 - rows in ASCENDING order - exchanges the counter for zero (ds_and_rtn): the first posting of a target that still
//     finds its count is the target's first touch (row, target), later ones find 0; the table is clean again when the pass ends;
//   * a first touch with count c >= 2 is a key [15 - c][row][target] as above.  Inside a pass the keys of one count value arrive in
//     ascending order, so the FIRST key of a count value in a partition is the partition's smallest for that value (all the prefix maxima
//     need: lane c keeps the minimum over the partitions); a key is kept only while it is below the K-th smallest kept so far (the kept
//     list is compacted to its K smallest whenever it fills, as in the CL instantiation) - no record list, nothing grows with the family.
// It takes its units from the CL instantiation's deferred list and hands the ones it cannot take (more than 15 sampled rows, a window that
// does not fit the chunk list) on to k_rank through UgsRank2Params::defer2.  udbusortedsearcherbig.cpp:82-110, countsort.cpp:110-191.
template <int D, bool CL, bool P16, bool HV = false>
__global__ __launch_bounds__(64, CL ? 3 : 4) void k_rank2(UgsDbView db, UgsBatchView bv, UgsRank2Params prm)
{
  static_assert(!(P16 && CL), "the cluster_fast instantiation reads the 32-bit postings (its index grows batch by batch)");
  static_assert(!P16 || R2_V2, "16-bit postings exist for the v2 ring only");
  static_assert(!HV || (CL && R2_V2 && !P16), "the heavy-unit instantiation is a variant of the cluster_fast one");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x;
  const uint32_t lane4 = lane * 4u;
  // ---- LDS carve (the bitmap sits at offset 0 so that a posting's word address needs no add)
  const uint32_t G = HV ? db.gsize : prm.G, np = HV ? db.np : prm.np, K = bv.K, kcap = prm.kcap;
  const uint32_t bm_bytes = HV ? G / 2u : G / 8u;                       // (HV: a 4-bit counter per target instead of one bit)
  const uint32_t *const ptab = HV ? db.part : db.part2;                 // the partition table that goes with G
  constexpr uint32_t SCAP = CL ? R2_SCAP_CL : R2_SCAP;                 // records of one partition the grouping handles
  uint32_t *s_stg = (uint32_t *)(smem + bm_bytes);                      // [SCAP] records of the partition being scanned
  uint32_t *s_hba = s_stg + SCAP;                                       // [R2_HB_BITS / 32] hash filter of the grouping
  uint32_t *s_c2 = s_hba + R2_HB_BITS / 32;                             // [16] kept count-2 keys per row
  uint32_t *s_cum = s_c2 + 16;                                          // [16] ... with that row or a lower one
  uint32_t *s_fpk = s_cum + 16;                                         // [16] smallest key per (15 - count)
  uint32_t *s_slots = s_fpk + 16;                                       // [16] sampled slots of the unit
  uint32_t *s_sel = s_stg;                                              // [64] selected targets for the fill (the staging area is idle by then)
  uint64_t *s_rs = (uint64_t *)(s_slots + 16);                            // [16] first posting (element index) of the unit's rows
  uint32_t *s_pe = (uint32_t *)(s_rs + 16);                             // [32] chunk index at which each partition of the window ends (W <= 28)
  uint2 *s_cl = (uint2 *)(s_stg + r2_fixed_words(CL));                  // [clcap] chunk list of the window being scanned (== s_pe + 32)
  const uint32_t clcap = prm.clcap;
  uint32_t *s_kl = (uint32_t *)(s_cl + clcap);                          // [kcap + 4] kept keys
  const uint32_t amax = bm_bytes - 4u;                                  // last word of the bitmap (G need not be a power of two)
  const uint32_t units = HV ? (uint32_t)bv.counters[UGS_CTR_DEFER] : bv.nq * bv.nstrand;      // (HV: the deferred list of the kernel in front)
  const uint32_t ns_max = prm.ns_max;
  const uint32_t *postings = db.postings;
  unsigned long long n_done_local = 0;
  if constexpr (HV) {                                                   // the counter table starts clean and every pass 2 leaves it clean
    for (uint32_t k = lane * 4u; k < bm_bytes; k += 256u) *(uint32_t *)(smem + k) = 0u;
  }
#ifdef R2_CLOCKS
  unsigned long long tc_pre = 0, tc_scan = 0, tc_fin = 0, tc_sel = 0;
#define R2_CLK(...) __VA_ARGS__
#else
#define R2_CLK(...)
#endif
#ifdef R2_CLOCKS2
#define R2_CLK2(...) __VA_ARGS__
#else
#define R2_CLK2(...)
#endif

  for (uint32_t k = lane; k < R2_HB_BITS / 32; k += 64) s_hba[k] = 0;         // the filter starts clean and is left clean

  // units are handed out four at a time (one same-address atomic per unit would be a tenth of this kernel) - except the last eight per
  // wave of the grid, which go one at a time: a launch of 125 k units gives a wave 30 of them, and whole fours at the end left some waves
  // a unit's work (a tenth of a small launch) behind the others
  const uint32_t coarse_end = units > 8u * gridDim.x ? units - 8u * gridDim.x : 0u;
  uint32_t ubase = 0, uidx = 4, ugot = 4;
  for (;;) {
    if (uidx == ugot) {
      const uint32_t want = (!HV && ubase + ugot < coarse_end) ? 4u : 1u;   // (ubase + ugot: the counter was at least there; heavy units singly)
      uint32_t v = 0;
      if (lane == 0) v = (uint32_t)atomicAdd(&bv.counters[HV ? UGS_CTR_NEXT_HEAVY : UGS_CTR_NEXT_RANK2], (unsigned long long)want);
      ubase = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
      uidx = 0; ugot = want;
    }
    uint32_t unit = ubase + uidx;
    ++uidx;
    if (unit >= units) break;
    if constexpr (HV) unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)bv.defer_list[unit]);
    else if constexpr (CL) { if (bv.unit_order) unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)bv.unit_order[unit]); }     // (heaviest first)
    const uint32_t ns = (uint32_t)__builtin_amdgcn_readfirstlane((int)bv.unit_ns[unit]);
    bool bad = ns > 15u;                                                  // 4-bit count field of the keys
    if constexpr (CL && !HV) { if (prm.force_defer) bad = true; }
#ifdef R2_DEFER_STATS
    if (bad && lane == 0) atomicAdd(&bv.counters[UGS_CTR_T4], 1ull);
#endif
    if (ns == 0) {
      if (lane == 0) { bv.cand_n[unit] = 0; if constexpr (CL) { bv.cl_info[(uint64_t)unit * 4 + 0] = 0; bv.cl_info[(uint64_t)unit * 4 + 1] = 0; bv.cl_info[(uint64_t)unit * 4 + 2] = 0; } }
      continue;
    }
    uint32_t nk = 0, n_stg = 0;
    bool any_posting = false;
    R2_CLK(const unsigned long long tk0 = clock64(); unsigned long long tfin = 0, tpre = 0;)
    if (!bad) {
      // ---- row descriptors
      {
        const bool rowlane = lane < ns;
        const uint32_t slot = rowlane ? bv.unit_slots[(uint64_t)unit * ns_max + lane] : 0u;
        const uint64_t rs = rowlane ? db.row_off[slot] : 0ull;            // the row's first posting (element index)
        if (lane < 16) { s_slots[lane] = slot; s_rs[lane] = rs; s_c2[lane] = 0; s_cum[lane] = 0; if constexpr (CL) s_fpk[lane] = R2_KEY_INF; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
      // one ds_or_rtn per posting; invalid postings OR a zero into an in-range word
      auto count_chunk = [&](uint32_t meta, R2Stage &S) {
        const int vlen = (int)R2D_N(meta) - (int)lane4;
        const uint32_t sub = R2D_PART(meta) * G;
        // (lanes beyond the chunk hold copies of lane 0's postings: an atomic of theirs, even one that ORs a zero, would go to the SAME
        // word as lane 0's and same-address atomics are served one after the other - they stay out of the LDS instructions altogether)
        const bool lane_on = vlen > 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { S.old[j] = 0; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t t = S.t[j];
          uint32_t ad = ((t - sub) >> 3) & ~3u;
          ad = ad < amax ? ad : amax;                                    // (only a posting behind the chunk's end can be out of range: it ORs a zero)
          const uint32_t bit = j < vlen ? (1u << (t & 31u)) : 0u;
          S.bit[j] = bit;
#if defined(R2_PROBE_NOATOM)                 // timing probes (tools/r2_probe.sh): wrong results
          S.old[j] = 0; if (ad == 0xffffffffu) S.old[j] = bit;
#elif defined(R2_PROBE_NOCONFLICT)           // every lane its own bank (wrong results)
          S.old[j] = __hip_atomic_fetch_or((lds32)(uintptr_t)((lane & 31u) * 4u + (ad & 0x1f80u)), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#elif defined(R2_PROBE_READONLY)             // (wrong results)
          S.old[j] = *(volatile lds32)(uintptr_t)ad;
#elif defined(R2_READ_OR)                    // plain read + non-returning or: the same result (only this wave touches its bitmap, a row holds a target once)
          S.old[j] = *(volatile lds32)(uintptr_t)ad;
          __hip_atomic_fetch_or((lds32)(uintptr_t)ad, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
          if (lane_on) S.old[j] = __hip_atomic_fetch_or((lds32)(uintptr_t)ad, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
        }
      };
      // postings that found their bit set leave a record (row << 24 | target); the chunk's postings are still in its ring slot
      auto emit_chunk = [&](const R2Stage &S, uint32_t meta) {
#if defined(R2_PROBE_NOEMIT)
        if ((S.old[0] & S.bit[0]) == 0x12345u) s_stg[0] = S.t[0];
        return;
#endif
        const uint32_t rtag = R2D_ROW(meta) << 24;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool hit = (S.old[j] & S.bit[j]) != 0u;
          const uint64_t m = r2_ballot(hit);
          if (m) {
            uint32_t pos = n_stg + r2_mbcnt(m);
            pos = pos < SCAP - 1u ? pos : SCAP - 1u;                   // (beyond the capacity the unit is deferred: n_stg tells)
            if (hit) s_stg[pos] = S.t[j] | rtag;
            n_stg += (uint32_t)__popcll(m);
          }
        }
      };
      auto zero_bitmap = [&]() {
        uint4 z; z.x = z.y = z.z = z.w = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // (bm_bytes is a multiple of 1024: every store is all lanes or none - straight-line stores with immediate offsets)
        unsigned char *zp = smem + lane * 16u;
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) if (k * 1024u < bm_bytes) *(uint4 *)(zp + k * 1024u) = z;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      };
      // ---- a partition is done: group its records by target, make keys, prune, keep (NB register batches of 64 records)
      // CL: the kept-key list is about to overflow - its K smallest keys stay (all-pairs ranks, keys are distinct), in rank order
      auto compact_kept = [&]() {
        constexpr int CB = 8;                                            // register batches of 64 keys (kcap <= 508 in this mode)
        uint32_t ck[CB], cr[CB];
        if (lane < 4u) s_kl[nk + lane] = R2_KEY_INF;                     // (padding for the 4-wide loop; nk <= kcap)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int q = 0; q < CB; ++q) {
          const uint32_t i = (uint32_t)q * 64u + lane;
          ck[q] = R2_KEY_INF; cr[q] = 0xffffffffu;
          if ((uint32_t)q * 64u < nk) {
            ck[q] = i < nk ? s_kl[i] : R2_KEY_INF;
            uint32_t rank = 0;
            const uint4 *k4 = (const uint4 *)s_kl;
            for (uint32_t j = 0; j < nk; j += 4u) {
              const uint4 x = k4[j >> 2];
              rank += (x.x < ck[q]) + (x.y < ck[q]) + (x.z < ck[q]) + (x.w < ck[q]);
            }
            cr[q] = rank;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int q = 0; q < CB; ++q) if (ck[q] != R2_KEY_INF && cr[q] < K) s_kl[cr[q]] = ck[q];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        nk = nk < K ? nk : K;
      };
      auto finalize_nb = [&](auto nbc, uint32_t n) {
        constexpr int NB = decltype(nbc)::value;
        if constexpr (CL) { if (nk + n > kcap) compact_kept(); }
        uint32_t rec[NB], t[NB], row[NB], wofs[NB], hbit[NB], cnt[NB], cumv[NB]; bool act[NB], fl[NB], drop[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const uint32_t i = (uint32_t)b * 64u + lane;
          act[b] = i < n;
          rec[b] = act[b] ? s_stg[i] : 0u;
          t[b] = rec[b] & 0xffffffu; row[b] = rec[b] >> 24;
          const uint32_t h = (t[b] ^ (t[b] >> 11)) & (R2_HB_BITS - 1u);
          wofs[b] = h >> 5; hbit[b] = 1u << (h & 31u);
          cnt[b] = 2; drop[b] = false;
          cumv[b] = act[b] ? s_cum[row[b]] : 0u;
        }
        // the hash filter: "a record with this hash came before me" flags every record but the first of a bucket; a flagged record's
        // target is then compared with ALL records (the first of the bucket included), so one pass over the filter is enough
        uint32_t olda[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) olda[b] = act[b] ? atomicOr(&s_hba[wofs[b]], hbit[b]) : 0u;
#pragma unroll
        for (int b = 0; b < NB; ++b) fl[b] = act[b] && (olda[b] & hbit[b]) != 0u;
#pragma unroll
        for (int b = 0; b < NB; ++b) if (act[b]) s_hba[wofs[b]] = 0;                              // the filter is clean again
        {
          // the flagged records are taken a TARGET at a time.  The records were staged in scan order = descending rows, so a target's
          // LAST record carries its first-touch row; count = its records + 1, every other record of it is dropped.  (cluster_fast:
          // most records share their target with others - the centroids of the read's species are hit by every sampled word)
          uint64_t fm[NB];
#pragma unroll
          for (int b = 0; b < NB; ++b) fm[b] = r2_ballot(fl[b]);
#pragma unroll
          for (int bb = 0; bb < NB; ++bb) {
            while (fm[bb]) {
              const int L = __ffsll((long long)fm[bb]) - 1;
              const uint32_t tL = (uint32_t)__builtin_amdgcn_readlane((int)t[bb], L);
              uint32_t nsame = 0; int lastb = 0; uint64_t lastm = 0;
              uint64_t mbs[NB];
#pragma unroll
              for (int b = 0; b < NB; ++b) {
                mbs[b] = r2_ballot(act[b] && t[b] == tL);
                nsame += (uint32_t)__popcll(mbs[b]);
                if (mbs[b]) { lastb = b; lastm = mbs[b]; }
                fm[b] &= ~mbs[b];
              }
              if (nsame >= 2u) {                                        // (1: another target in the same bucket)
                const uint32_t lastl = 63u - (uint32_t)__builtin_clzll(lastm);
#pragma unroll
                for (int b = 0; b < NB; ++b) if (act[b] && t[b] == tL) { cnt[b] = nsame + 1u; drop[b] = !(b == lastb && lane == lastl); }
              }
            }
          }
        }
        // keys; a count-2 key stays only while fewer than K count-2 keys of EARLIER partitions with its row or a lower one are kept
        uint32_t nk_new = nk;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const uint32_t key = ((15u - cnt[b]) << 28) | rec[b];
          if constexpr (CL) { if (act[b] && !drop[b]) atomicMin(&s_fpk[15u - cnt[b]], key); }      // (every target's key, kept or not)
          const bool keep = act[b] && !drop[b] && (cnt[b] >= 3u || cumv[b] < K);
          const uint64_t m = r2_ballot(keep);
          uint32_t pos = nk_new + r2_mbcnt(m);
          pos = pos < kcap ? pos : kcap;                                 // (slot kcap is scratch; nk > kcap defers the unit)
          if (keep) s_kl[pos] = key;
          if (keep && cnt[b] == 2u) atomicAdd(&s_c2[row[b]], 1u);
          nk_new += (uint32_t)__popcll(m);
        }
        nk = nk_new;
        if (nk > kcap) { bad = true; return; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        {
          const uint32_t c = lane < 16u ? s_c2[lane] : 0u;
          const uint32_t incl = r2_row16_incl_sum(c);
          if (lane < 16u) s_cum[lane] = incl;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      };
      auto finalize = [&]() {
        const uint32_t n = n_stg;
        n_stg = 0;
        if (n == 0) return;
#if defined(R2_PROBE_NOFIN)
        return;
#endif
#ifdef R2_DEFER_STATS      // (tuning build: why units are deferred - two 32-bit counts per word: T4 rows > 15 | chunk list, T5 .. T7 records of the overflowing partition: <= 256 | <= 512, <= 1024 | <= 2048, <= 4096 | more)
        if (n > SCAP && lane == 0) { uint32_t bkt = 0; while (bkt < 5u && (256u << bkt) < n) ++bkt; atomicAdd(&bv.counters[UGS_CTR_T5 + (bkt >> 1)], 1ull << (32u * (bkt & 1u))); }
#endif
        if (n > SCAP) { bad = true; return; }
        if constexpr (CL) { if (n > 128u) { finalize_nb(std::integral_constant<int, 4>{}, n); return; } }
        if (n > 64u) finalize_nb(std::integral_constant<int, 2>{}, n); else finalize_nb(std::integral_constant<int, 1>{}, n);
      };

      // ---- the scan, a window of W partitions at a time
      // (as many partitions per window as the chunk list holds for this unit's row count: a multiple of 4)
      uint32_t W = ((clcap - 16u) / ((ns + 4u) & ~3u)) & ~3u;
      W = W < 4u ? 4u : (W > prm.W ? prm.W : W);
      // HV: W counts VIRTUAL partitions - every partition appears twice in the chunk list, pass 1 then pass 2 (W / 2 partitions per window)
      uint32_t hv_seen = 0, hv_kth = R2_KEY_INF;                         // count values met in the running pass 2 | the K-th smallest kept key
      uint32_t hv_fk = R2_KEY_INF;                                       // lane c: smallest key of count c over the partitions scanned
      (void)hv_seen; (void)hv_kth; (void)hv_fk;
      R2_CLK(tpre = clock64() - tk0;)                                     // (the unit's prologue: row descriptors)
      for (uint32_t p0 = 0; p0 < np && !bad; ) {
        const uint32_t Wn = HV ? ((np - p0) * 2u < W ? (np - p0) * 2u : W) : (np - p0 < W ? np - p0 : W);      // (HV: virtual partitions)
        R2_CLK(const unsigned long long tw0 = clock64();)
        // (1) the window's chunk list -> LDS.  Lane = (partition of the window: lane >> 4, row: ns - 1 - (lane & 15)), so lane order is the
        // scan order (partition ascending, row DESCENDING); a sub-row's bounds are two adjacent words of the row's partition-table
        // line; a sub-row longer than 256 postings gives several chunks.  Inside the scan nothing but the posting loads touches
        // global memory.
        // Every partition's chunks are padded to a multiple of D with empty descriptors: a partition then always ends where the ring's
        // loop body ends, and its grouping has one call site instead of one per ring stage.
        uint32_t nch = 0;
        for (uint32_t pl0 = 0; pl0 < Wn; pl0 += 4u) {
          const uint32_t grp = lane >> 4, pl = pl0 + grp, rr = lane & 15u;
          const bool valid = rr < ns && pl < Wn;
          // (HV: virtual partition pl = partition pl / 2, pass pl % 2; rows in ASCENDING order - the first posting that still finds its
          //  target's count in pass 2 is the target's first touch)
          const uint32_t r = valid ? (HV ? rr : ns - 1u - rr) : 0u, p = HV ? p0 + (pl >> 1) : p0 + pl;
          u32x2 lh; lh.x = 0; lh.y = 0;
          if (valid) __builtin_memcpy(&lh, ptab + (uint64_t)s_slots[r] * (np + 1u) + p, 8);
          const uint32_t len = lh.y - lh.x;
          const uint64_t a0 = s_rs[r] + lh.x;
          // (P16: chunks start at multiples of four elements; the sub-row's first chunk begins up to three elements early)
          const uint32_t lead = P16 ? ((uint32_t)a0 & 3u) : 0u;
          const uint32_t spanall = len ? lead + len : 0u;
          const uint32_t nc = (spanall + 255u) >> 8;
          const uint32_t incl = r2_row16_incl_sum(nc);                    // inside the partition's 16 lanes
          uint32_t gbase = nch, mybase = 0, mytot = 0, mypad = 0;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, g * 16 + 15);
            const uint32_t pad = (tot + (uint32_t)D - 1u) / (uint32_t)D * (uint32_t)D;
            if (grp == (uint32_t)g) { mybase = gbase; mytot = tot; mypad = pad; }
            gbase += pad;
          }
          const uint32_t base = mybase + incl - nc;
          for (uint32_t j = 0; r2_ballot(j < nc) != 0ull; ++j) {
            if (j < nc && base + j < clcap) {
              const uint64_t a = a0 - lead + (uint64_t)j * 256u;
              const uint32_t n = spanall - j * 256u < 256u ? spanall - j * 256u : 256u;
              uint2 e; e.x = (uint32_t)a; e.y = n | (r << 9) | (p << 13) | ((uint32_t)(a >> 32) << 24);
              if constexpr (P16) { if (j == 0u) e.y |= lead << 22; }
              if constexpr (HV) e.y |= (pl & 1u) << 23;
              s_cl[base + j] = e;
            }
          }
          if (rr < mypad - mytot && mybase + mytot + rr < clcap) { uint2 e; e.x = 0; e.y = 0; s_cl[mybase + mytot + rr] = e; }
          if (rr == 15u && pl < 32u) s_pe[pl] = mybase + mypad;           // list index at which this partition ends
          nch = gbase;
        }
        if (nch == 0) { p0 += HV ? Wn >> 1 : Wn; continue; }
        any_posting = true;
        R2_CLK(tpre += clock64() - tw0;)
        // padding entries: the ring below issues exactly one load per stage (its counted waits depend on it) and looks D chunks ahead
#ifdef R2_DEFER_STATS
        if (nch + (R2_V2 ? 3u : 2u) * (uint32_t)D > clcap && lane == 0) atomicAdd(&bv.counters[UGS_CTR_T4], 1ull << 32);
#endif
        constexpr uint32_t PADN = (R2_V2 ? 3u : 2u) * (uint32_t)D;     // empty descriptors behind the list (v2 reads a loop body further ahead)
        if (nch + PADN > clcap) {                                      // a window with more chunks than the list holds: very long rows
          if constexpr (CL) { if (W > (HV ? 2u : 1u)) { W = W > 4u ? 4u : W >> 1; continue; } }     // (cluster_fast: the window is cut down to one partition before the unit is given up)
          bad = true; break;
        }
        for (uint32_t i = nch + lane; i < nch + PADN; i += 64u) { uint2 e; e.x = 0; e.y = 0; s_cl[i] = e; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // (2) the ring: D = 4 chunk slots, three posting loads in flight behind the chunk being counted (r2_issue / r2_take above).  A
        // slot is waited for with vmcnt(D - 1): loads complete in order and every stage issues exactly one, so D - 1 younger ones are
        // in flight when a chunk's data is needed.  Stage k (chunk c + k): the records of the previous chunk are emitted, slot k is
        // taken into registers and loaded again with the chunk D ahead, then the chunk is counted.
#if R2_V2
        // v2 of the ring (r5).  What changed against v1 (kept below for A/B builds, -DR2_V2=0):
        //  * the descriptors of a whole loop body (four chunks) are fetched by TWO broadcast reads, before the body's last chunk is
        //    counted - v1 read one descriptor per stage right behind the stage's atomics, and the read's lgkmcnt(0) made every stage
        //    wait out its own four ds_or_rtn;
        //  * a chunk's records are emitted one stage LATER, behind the next chunk's posting wait: the atomics' round trip lies in
        //    the shadow of s_waitcnt vmcnt; the postings of two chunks are live for it (four more registers);
        //  * the partition's first target (sub) is a loop-carried scalar set where a partition starts, not decoded per chunk;
        //  * postings behind a chunk's end OR a zero wherever they point (no address clamp), returned words are never initialised
        //    (bit == 0 masks them), a padding chunk counts and emits nothing by its length alone (no per-stage scalar test).
        static_assert(D == 4, "the ring is written for four slots");
        uint32_t mt[4], dlo[4], dhi[4];                                  // metas of the chunks in the ring | descriptors of the next four
        auto fetch4 = [&](uint32_t j) {
          const uint4 a = *(const uint4 *)(s_cl + j), b4 = *(const uint4 *)(s_cl + j + 2u);     // (uniform addresses: LDS broadcast reads)
          dlo[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.x); dhi[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.y);
          dlo[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.z); dhi[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.w);
          dlo[2] = (uint32_t)__builtin_amdgcn_readfirstlane((int)b4.x); dhi[2] = (uint32_t)__builtin_amdgcn_readfirstlane((int)b4.y);
          dlo[3] = (uint32_t)__builtin_amdgcn_readfirstlane((int)b4.z); dhi[3] = (uint32_t)__builtin_amdgcn_readfirstlane((int)b4.w);
        };
        auto src_of = [&](int k, uint32_t &voff) -> const uint32_t * {
          // lane l reads postings [4l, 4l+4) of the chunk; lanes beyond it re-read the chunk's first 16 bytes (no traffic) and are masked
          if constexpr (P16) {
            voff = lane4 < R2D_N(dhi[k]) ? lane4 * 2u : 0u;               // (four 16-bit postings = 8 bytes per lane)
            return (const uint32_t *)(prm.post16 + (((uint64_t)R2D_AHI(dhi[k]) << 32) | dlo[k]));
          }
          voff = lane4 < R2D_N(dhi[k]) ? lane4 * 4u : 0u;
          return postings + (((uint64_t)R2D_AHI(dhi[k]) << 32) | dlo[k]);
        };
        uint32_t nsub8 = 0;                                              // minus (first target of the partition being scanned) / 8
        uint32_t o0, o1, o2, o3;                                         // what the atomics of the chunk in flight returned
        asm volatile("" : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3));       // (never read where the chunk's bit is 0)
        uint32_t e_bit[4] = {0, 0, 0, 0}, e_t[4] = {0, 0, 0, 0}, e_meta = 0;
        auto count2 = [&](uint32_t meta, const uint32_t (&t)[4]) {
          const int vlen = (int)R2D_N(meta) - (int)lane4;
          if constexpr (HV) {
            // 4-bit counters: target t of partition p lives in nibble (t - p G) % 8 of word (t - p G) / 8
            const uint32_t sub = R2D_PART_HV(meta) * G;
            const bool second = R2D_PASS2_HV(meta) != 0u;
            e_meta = meta;
#pragma unroll
            for (int j = 0; j < 4; ++j) { e_bit[j] = 0u; e_t[j] = t[j]; }
            if (vlen > 0) {
              uint32_t ad[4], sh[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const uint32_t idx = t[j] - sub;
                const bool ok = j < vlen;
                ad[j] = ok ? (idx >> 3) << 2 : 0u;
                sh[j] = (idx & 7u) << 2;
                e_bit[j] = (second && ok) ? (0x80000000u | sh[j]) : 0u;  // (what the emission needs: "valid" and the nibble's shift)
              }
              if (!second) {
#pragma unroll
                for (int j = 0; j < 4; ++j) (void)__hip_atomic_fetch_add((lds32)(uintptr_t)ad[j], j < vlen ? (1u << sh[j]) : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              } else {
                o0 = __hip_atomic_fetch_and((lds32)(uintptr_t)ad[0], 0 < vlen ? ~(15u << sh[0]) : 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                o1 = __hip_atomic_fetch_and((lds32)(uintptr_t)ad[1], 1 < vlen ? ~(15u << sh[1]) : 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                o2 = __hip_atomic_fetch_and((lds32)(uintptr_t)ad[2], 2 < vlen ? ~(15u << sh[2]) : 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                o3 = __hip_atomic_fetch_and((lds32)(uintptr_t)ad[3], 3 < vlen ? ~(15u << sh[3]) : 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            }
            return;
          }
          if constexpr (P16) {
            // t[0], t[1] = four 16-bit postings (offsets inside the partition).  Validity: element 4 lane + j of the chunk counts when
            // lead <= 4 lane + j < span - the lead only ever concerns lane 0
            const uint32_t x0 = t[0], x1 = t[1];
            const uint32_t hi_n = (uint32_t)(vlen < 0 ? 0 : (vlen > 4 ? 4 : vlen));
            uint32_t m4 = (1u << hi_n) - 1u;
            m4 &= 0xfu << (lane == 0u ? R2D_LEAD16(meta) : 0u);
            e_t[0] = x0; e_t[1] = x1;
            e_bit[0] = (m4 & 1u) << (x0 & 31u);
            e_bit[1] = ((m4 >> 1) & 1u) << ((x0 >> 16) & 31u);
            e_bit[2] = ((m4 >> 2) & 1u) << (x1 & 31u);
            e_bit[3] = ((m4 >> 3) & 1u) << ((x1 >> 16) & 31u);
            e_meta = meta;
            if (vlen > 0) {
              // bitmap word of offset o: byte (o >> 5) * 4 - the bitmap sits at LDS offset 0 and o < G <= 65536
              o0 = __hip_atomic_fetch_or((lds32)(uintptr_t)((x0 >> 3) & 0x1ffcu), e_bit[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              o1 = __hip_atomic_fetch_or((lds32)(uintptr_t)((x0 >> 19) & 0x1ffcu), e_bit[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              o2 = __hip_atomic_fetch_or((lds32)(uintptr_t)((x1 >> 3) & 0x1ffcu), e_bit[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              o3 = __hip_atomic_fetch_or((lds32)(uintptr_t)((x1 >> 19) & 0x1ffcu), e_bit[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            return;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) { e_bit[j] = j < vlen ? (1u << (t[j] & 31u)) : 0u; e_t[j] = t[j]; }
          e_meta = meta;
          if (vlen > 0) {
            // (lanes beyond the chunk hold copies of lane 0's postings: they stay out of the LDS instructions - same-address atomics are
            // served one after the other; a posting behind the chunk's end inside an active lane ORs a zero, wherever it points)
            // word address = ((t - sub) >> 5) * 4 = ((t >> 5) << 2) - sub / 8 (sub is a multiple of 8192): a shift and one v_lshl_add
            o0 = __hip_atomic_fetch_or((lds32)(uintptr_t)r2_word_addr(t[0], nsub8), e_bit[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            o1 = __hip_atomic_fetch_or((lds32)(uintptr_t)r2_word_addr(t[1], nsub8), e_bit[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            o2 = __hip_atomic_fetch_or((lds32)(uintptr_t)r2_word_addr(t[2], nsub8), e_bit[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            o3 = __hip_atomic_fetch_or((lds32)(uintptr_t)r2_word_addr(t[3], nsub8), e_bit[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        };
        // the records of the chunk counted one stage ago: postings that found their bit set (row << 24 | target)
        auto emit2 = [&]() {
          const uint32_t rtag = R2D_ROW(e_meta) << 24;
          const uint32_t old[4] = {o0, o1, o2, o3};
          if constexpr (HV) {
            // the first touches of the pass-2 chunk counted one stage ago: postings that still found their target's count
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const bool on = (e_bit[j] & 0x80000000u) != 0u;
              const uint32_t c = on ? (old[j] >> (e_bit[j] & 31u)) & 15u : 0u;
              const uint32_t key = ((15u - c) << 28) | rtag | e_t[j];
              const bool cand = c >= 2u;
              uint64_t mc = r2_ballot(cand);
              if (mc) {
                // (1) the first key of a count value in this pass is the partition's smallest for it (keys ascend inside a pass)
                uint64_t mn = r2_ballot(cand && ((hv_seen >> c) & 1u) == 0u);
                while (mn) {
                  const int L = __ffsll((long long)mn) - 1;
                  const uint32_t cL = (uint32_t)__builtin_amdgcn_readlane((int)c, L), kL = (uint32_t)__builtin_amdgcn_readlane((int)key, L);
                  if (lane == cL) hv_fk = kL < hv_fk ? kL : hv_fk;
                  hv_seen |= 1u << cL;
                  mn &= ~r2_ballot(c == cL);
                }
                // (2) a key stays while it is below the K-th smallest kept so far
                uint64_t mk = r2_ballot(cand && key < hv_kth);
                if (mk) {
                  if (nk + (uint32_t)__popcll(mk) > kcap) {
                    compact_kept();                                     // (nk = min(nk, K), the kept keys in rank order)
                    hv_kth = nk >= K ? (uint32_t)__builtin_amdgcn_readfirstlane((int)s_kl[K - 1u]) : R2_KEY_INF;
                    mk = r2_ballot(cand && key < hv_kth);
                  }
                  if (mk) {
                    const uint32_t pos = nk + r2_mbcnt(mk);
                    if (cand && key < hv_kth) s_kl[pos] = key;
                    nk += (uint32_t)__popcll(mk);
                  }
                }
              }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) e_bit[j] = 0;
            return;
          }
          const uint32_t psub = P16 ? R2D_PART16(e_meta) * G : 0u;        // (P16: the partition's first target, added back to the offsets)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool hit = (old[j] & e_bit[j]) != 0u;
            const uint64_t m = r2_ballot(hit);
            if (m) {
              uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, n_stg));   // n_stg + rank
              pos = pos < SCAP - 1u ? pos : SCAP - 1u;                 // (beyond the capacity the unit is deferred: n_stg tells)
              if constexpr (P16) {
                const uint32_t x = e_t[j >> 1];
                const uint32_t o16 = (j & 1) ? (x >> 16) : (x & 0xffffu);
                if (hit) s_stg[pos] = (psub + o16) | rtag;
              } else {
                if (hit) s_stg[pos] = e_t[j] | rtag;
              }
              n_stg += (uint32_t)__popcll(m);
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) e_bit[j] = 0;                      // (emitted: a second call finds nothing)
        };
        fetch4(0);
        { uint32_t vo; const uint32_t *sp;
          sp = src_of(0, vo); r2_issue<0, P16>(vo, sp); sp = src_of(1, vo); r2_issue<1, P16>(vo, sp);
          sp = src_of(2, vo); r2_issue<2, P16>(vo, sp); sp = src_of(3, vo); r2_issue<3, P16>(vo, sp); }
#pragma unroll
        for (int k = 0; k < 4; ++k) mt[k] = dhi[k];
        fetch4(4);
        uint32_t pi = 0, pend = 0, pe_next = s_pe[0];
#define R2_STAGE2(k, last)                                                                                       \
          {                                                                                                      \
            uint32_t tn[4], vo;                                                                                  \
            const uint32_t *sp = src_of(k, vo);                                                                  \
            r2_take<k, P16>(tn);                                                                                 \
            r2_issue<k, P16>(vo, sp);                                                                            \
            emit2();                                                                                             \
            const uint32_t pm = mt[k]; mt[k] = dhi[k];                                                           \
            if (last) fetch4(c + 2u * (uint32_t)D);                                                              \
            count2(pm, tn);                                                                                      \
          }
        for (uint32_t c = 0; c < nch && !bad; c += (uint32_t)D) {
          if (c == pend) {
            // a partition ended with the previous loop body: its last chunk's records, then its grouping
            R2_CLK(const unsigned long long tf0 = clock64();)
            emit2();
            if constexpr (!HV) { if (c != 0) { finalize(); if (bad) break; } }
            // (the next partition's end index was read where this one began: no LDS round trip at the boundary)
            do { pend = (uint32_t)__builtin_amdgcn_readfirstlane((int)pe_next); ++pi; pe_next = s_pe[pi]; } while (pend == c);      // (partitions without a posting)
            if constexpr (HV) hv_seen = 0u;                               // (a pass begins: pass 2 left the counter table clean, nothing to zero)
            else {
            if constexpr (!P16) nsub8 = 0u - R2D_PART(mt[0]) * (G >> 3);  // (a partition's first chunk is never padding)
            zero_bitmap();
            }
            R2_CLK(tfin += clock64() - tf0;)
          }
          R2_STAGE2(0, false) R2_STAGE2(1, false) R2_STAGE2(2, false) R2_STAGE2(3, true)
        }
#undef R2_STAGE2
        if (!bad) {
          R2_CLK(const unsigned long long tf0 = clock64();)
          emit2();
          if constexpr (!HV) finalize();
          R2_CLK(tfin += clock64() - tf0;)
        }
#else
        uint32_t mt[D];
        auto fetch = [&](uint32_t j, uint32_t &meta, uint32_t &voff) -> const uint32_t * {
          const uint2 e = s_cl[j];                                       // (uniform address: an LDS broadcast read)
          const uint32_t elo = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x), ehi = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y);
          meta = ehi;
          const uint64_t a = ((uint64_t)R2D_AHI(ehi) << 32) | elo;
          // lane l reads postings [4l, 4l+4) of the chunk; lanes beyond it re-read the chunk's first 16 bytes (no traffic) and are masked
          voff = lane4 < R2D_N(ehi) ? lane4 * 4u : 0u;
          return postings + a;                                           // (scalar add: the base register pair is written by the scalar unit)
        };
        { uint32_t vo; const uint32_t *sp;
          sp = fetch(0, mt[0], vo); r2_issue<0>(vo, sp); sp = fetch(1, mt[1], vo); r2_issue<1>(vo, sp);
          sp = fetch(2, mt[2], vo); r2_issue<2>(vo, sp); sp = fetch(3, mt[3], vo); r2_issue<3>(vo, sp); }
        static_assert(D == 4, "the ring is written for four slots");
        R2Stage S; uint32_t pm = 0; bool pv = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) { S.old[j] = 0; S.bit[j] = 0; S.t[j] = 0; }
        uint32_t pi = 0, pend = 0;
#if defined(R2_PROBE_NOCOUNT)
#define R2_COUNT S.old[0] = S.t[0] ^ S.t[1] ^ S.t[2] ^ S.t[3]; S.bit[0] = 0x40000000u; pv = true;
#else
#define R2_COUNT count_chunk(pm, S); pv = true;
#endif
#define R2_STAGE(k)                                                                                              \
          {                                                                                                      \
            uint32_t nm, vo;                                                                                     \
            const uint32_t *sp = fetch(c + (uint32_t)(k) + (uint32_t)D, nm, vo);                                 \
            if (pv) { emit_chunk(S, pm); pv = false; }                                                           \
            r2_take<k>(S.t);                                                                                     \
            r2_issue<k>(vo, sp);                                                                                 \
            pm = mt[k]; mt[k] = nm;                                                                              \
            if (R2D_N(pm) != 0u) { R2_COUNT }                 /* (padding descriptors are not counted) */         \
          }
        for (uint32_t c = 0; c < nch && !bad; c += (uint32_t)D) {
          // the last chunk of the previous loop body, then - where a partition ended there - its grouping
          if (pv) { emit_chunk(S, pm); pv = false; }
          if (c == pend) {
            if (c != 0) { finalize(); if (bad) break; }
            do { pend = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_pe[pi]); ++pi; } while (pend == c);      // (partitions without a posting)
            zero_bitmap();
          }
          R2_STAGE(0) R2_STAGE(1) R2_STAGE(2) R2_STAGE(3)
        }
#undef R2_STAGE
#undef R2_COUNT
        if (!bad) {
          if (pv) { emit_chunk(S, pm); pv = false; }
          finalize();
        }
#endif
        // every load of the ring has landed before the next window (or unit) issues into the same slots
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        p0 += HV ? Wn >> 1 : Wn;
      }
      if constexpr (HV) {
        // the smallest key of every count value, as the CL instantiation's groupings leave it in s_fpk (index 15 - count)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane >= 2u && lane <= 15u) s_fpk[15u - lane] = hv_fk;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
    if (bad) {
      // outside this kernel's envelope: the general kernel takes the unit (it writes cand / cand_cnt / cand_n)
      if (lane == 0) {
        if constexpr (HV) {
          const unsigned long long idx = atomicAdd(&bv.counters[UGS_CTR_DEFER2], 1ull);
          prm.defer2[idx] = unit;
        } else {
          const unsigned long long idx = atomicAdd(&bv.counters[UGS_CTR_DEFER], 1ull);
          bv.defer_list[idx] = unit;
        }
      }
      if constexpr (HV) {                                                 // (a pass 1 may have been cut short: the table must be clean for the next unit)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (uint32_t k = lane * 4u; k < bm_bytes; k += 256u) *(uint32_t *)(smem + k) = 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
      continue;
    }
    ++n_done_local;
#if defined(R2_PROBE_NOSEL)
    if (nk != 0x7fffffffu) { if (lane == 0) bv.cand_n[unit] = 0; continue; }
#endif
    R2_CLK(const unsigned long long tk1 = clock64();)
    // ---- cut-offs from the smallest key of every count value (countsort.cpp:13-24,114-126)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if constexpr (!CL) { if (lane < 16u) s_fpk[lane] = R2_KEY_INF; }
    if (lane < 4u) s_kl[nk + lane] = R2_KEY_INF;                         // (padding for the 4-wide ranking loop; nk <= kcap)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if constexpr (!CL) for (uint32_t i = lane; i < nk; i += 64u) { const uint32_t key = s_kl[i]; atomicMin(&s_fpk[key >> 28], key); }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint32_t M = 0, nv = 0;
    {
      // lane c (2..15) looks at count c
      const uint32_t f = (lane >= 2u && lane <= 15u) ? s_fpk[15u - lane] : R2_KEY_INF;
      const uint64_t vm = r2_ballot(f != R2_KEY_INF);
      if (vm) {
        M = 63u - (uint32_t)__builtin_clzll(vm);
        const uint32_t fM = (uint32_t)__builtin_amdgcn_readlane((int)f, (int)M) & 0x0fffffffu;
        const uint64_t lm = r2_ballot(f != R2_KEY_INF && lane < M && (f & 0x0fffffffu) < fM);
        // (a count-1 first position below fp[M] gives NextValue 1 when no higher count qualifies: MinValue = NextValue / 2 is 0 either way)
        nv = lm ? 63u - (uint32_t)__builtin_clzll(lm) : 0u;
      } else M = any_posting ? 1u : 0u;
    }
    const uint32_t min_value = CL ? 0u : nv / 2u;                         // (cluster_fast: the host applies the cut-off of the merged scan)
    if constexpr (CL) {
      // the strict prefix maxima of the scan, as k_rank's cluster mode writes them: for c = M .. 1 the first position of count c where it
      // lies before the first position of every higher count.  Counts >= 2 come from s_fpk; a count-1 target is only ever an event
      // (and only ever decides NextValue) as the very FIRST posting of the scan: every posting in front of the first count >= 2 target
      // has count 1, and if there is none the first count-1 position lies behind that target
      uint64_t P0 = ~0ull;
      {
        const bool rowlane = lane < ns;
        const uint32_t slot = rowlane ? s_slots[lane] : 0u;
        const uint64_t ra = rowlane ? db.row_off[slot] : 0ull, rb = rowlane ? db.row_off[slot + 1] : 0ull;
        const uint64_t ne_rows = r2_ballot(rowlane && rb > ra);
        if (ne_rows) {
          const uint32_t r0 = (uint32_t)__ffsll((long long)ne_rows) - 1u;
          const uint64_t a0 = r2_readlane64(ra, r0);
          P0 = ((uint64_t)r0 << 32) | postings[a0];
        }
      }
      if (lane == 0) {
        uint32_t ne = 0, nv_out = nv;
        uint64_t sufmin = ~0ull;
        if (M >= 2u) {
          for (uint32_t c = M; c >= 2u; --c) {
            const uint32_t k32 = s_fpk[15u - c];
            if (k32 == R2_KEY_INF) continue;
            const uint64_t f = ((uint64_t)((k32 >> 24) & 15u) << 32) | (k32 & 0xffffffu);
            if (f < sufmin) { if (ne < UGS_CL_EV) bv.cl_ev[(uint64_t)unit * UGS_CL_EV + ne] = ((uint64_t)c << R2_POS_BITS) | f; ++ne; sufmin = f; }
          }
          if (nv_out < 2u) nv_out = P0 < sufmin ? 1u : 0u;
        } else nv_out = 0;
        if (M >= 1u && P0 < sufmin) { if (ne < UGS_CL_EV) bv.cl_ev[(uint64_t)unit * UGS_CL_EV + ne] = (1ull << R2_POS_BITS) | P0; ++ne; }
        bv.cl_info[(uint64_t)unit * 4 + 0] = M; bv.cl_info[(uint64_t)unit * 4 + 1] = nv_out; bv.cl_info[(uint64_t)unit * 4 + 2] = ne;
      }
    }
    const uint32_t cmin = min_value > 2u ? min_value : 2u;
    const uint32_t limit = (16u - cmin) << 28;                            // keys below it have count >= cmin
    // ---- rank the kept keys all-pairs (nk is small: ~K + the targets with count >= 3)
    uint32_t nsel = 0;
    {
      uint32_t nelig = 0;
      for (uint32_t e0 = 0; e0 < nk; e0 += 64u) {
        const uint32_t i = e0 + lane;
        const uint32_t key = i < nk ? s_kl[i] : R2_KEY_INF;
        const bool elig = key < limit;
        nelig += (uint32_t)__popcll(r2_ballot(elig));
        uint32_t rank = 0;
        const uint4 *k4 = (const uint4 *)s_kl;
        for (uint32_t j = 0; j < nk; j += 4u) {
          const uint4 x = k4[j >> 2];
          rank += (x.x < key) + (x.y < key) + (x.z < key) + (x.w < key);
        }
        if (elig && rank < K) {
          const uint32_t tg = key & 0xffffffu;
          bv.cand[(uint64_t)unit * K + rank] = tg;
          bv.cand_cnt[(uint64_t)unit * K + rank] = 15u - (key >> 28);
          if constexpr (CL) bv.cand_key[(uint64_t)unit * K + rank] = ((uint64_t)(R2_CMAXV - (15u - (key >> 28))) << R2_POS_BITS) | ((uint64_t)((key >> 24) & 15u) << 32) | tg;
          s_sel[rank] = tg;
        }
      }
      nsel = nelig < K ? nelig : K;
    }
    // ---- fewer than K targets with count >= 2 and the cut-off keeps count-1 targets: they follow in first-touch order = the postings
    // of row 0 ascending, then those of row 1 that no earlier row holds, ... - a target has count 1 exactly when it is not selected
    // (every count >= 2 target is, the list being exhausted): udbusortedsearcherbig.cpp:82-100 order, countsort.cpp:110-191
    if (nsel < K && min_value <= 1u && M >= 1u) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const uint32_t mine = lane < nsel ? s_sel[lane] : 0xffffffffu;
      uint32_t filled = nsel;
      for (uint32_t r = 0; r < ns && filled < K; ++r) {
        const uint32_t slot = s_slots[r];
        const uint64_t ra = db.row_off[slot], rb = db.row_off[slot + 1];
        for (uint64_t k0 = ra; k0 < rb && filled < K; k0 += 64) {
          const bool on = k0 + (uint64_t)lane < rb;
          const uint32_t t = on ? postings[k0 + lane] : 0u;
          bool in_set = false;
          for (uint32_t j = 0; j < nsel; ++j) in_set = in_set || t == (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)j);
          const bool e = on && !in_set;
          const uint64_t m = r2_ballot(e);
          const uint32_t rank = r2_mbcnt(m);
          if (e && filled + rank < K) {
            bv.cand[(uint64_t)unit * K + filled + rank] = t; bv.cand_cnt[(uint64_t)unit * K + filled + rank] = 1u;
            if constexpr (CL) bv.cand_key[(uint64_t)unit * K + filled + rank] = ((uint64_t)(R2_CMAXV - 1u) << R2_POS_BITS) | ((uint64_t)r << 32) | t;
          }
          const uint32_t n = (uint32_t)__popcll(m);
          filled = filled + n < K ? filled + n : K;
        }
      }
      nsel = filled;
    }
    if (lane == 0) bv.cand_n[unit] = nsel;
    R2_CLK(tc_pre += tpre; tc_fin += tfin; tc_scan += tk1 - tk0 - tpre - tfin; tc_sel += clock64() - tk1;)
  }
#ifdef R2_CLOCKS
  if (lane == 0) { atomicAdd(&bv.counters[UGS_CTR_T0], tc_pre); atomicAdd(&bv.counters[UGS_CTR_T1], tc_scan); atomicAdd(&bv.counters[UGS_CTR_T2], tc_fin); atomicAdd(&bv.counters[UGS_CTR_T3], tc_sel); }
#endif
  if (lane == 0 && n_done_local) atomicAdd(&bv.counters[HV ? UGS_CTR_HV_DONE : UGS_CTR_R2_DONE], n_done_local);
}

// ================================================================================================================================
// k_rank2g - the same kernel for SPARSE indexes (protein dictionaries: rows of a few hundred postings, 16-63 sampled rows per query).
// A (row, partition) sub-row holds a handful of postings there, so one chunk = the sub-rows of ALL sampled rows of a partition:
// every lane has its own descriptor (four consecutive postings of one row's sub-row: a 16-byte gather load), found per chunk from the
// rows' sub-row lengths (a prefix sum over the row lanes, a scatter of owner tags + a DPP max fill, three ds_bpermute).  Rows are laid
// out in DESCENDING order over the lanes and over a partition's chunks, so an earlier chunk never holds a lower row; inside a chunk
// the first touch of a target may be any of its lanes, so a record does not carry its own row but the LOWEST row among the chunk's
// postings of that target (the last matching lane).  Counts up to 63 and 64 rows do not fit the 32-bit keys: the kept keys are 64-bit
// ((255 - count) << 32 | row << 24 | target).  udbusortedsearcherbig.cpp:82-110, countsort.cpp:110-191 as k_rank2.
// ================================================================================================================================
#define R2G_HB_BITS 512u          // bits of each hash filter of the grouping (a handful of records per partition)
#define R2G_DCAP 256u            // descriptor lanes (quads of postings) of one partition; more: the unit is deferred

__global__ __launch_bounds__(64, 2) void k_rank2g(UgsDbView db, UgsBatchView bv, UgsRank2Params prm)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x;
  const uint32_t G = prm.G, np = prm.np, K = bv.K, kcap = prm.kcap;
  const uint32_t bm_bytes = G / 8u;
  // ---- LDS carve
  uint32_t *s_stg = (uint32_t *)(smem + bm_bytes);                      // [R2_SCAP + 64] records of the partition being scanned (+ slack)
  uint32_t *s_sel = s_stg;                                              //   after the scan: [64] selected targets (for the fill)
  uint64_t *s_fpk = (uint64_t *)(s_stg + 64);                           //   after the scan: [64] smallest key per count value
  uint32_t *s_hba = s_stg + R2_SCAP + 64;                               // [R2G_HB_BITS / 32] hash filter of the grouping
  uint32_t *s_c2 = s_hba + R2G_HB_BITS / 32;                            // [64] kept count-2 keys per row
  uint32_t *s_cum = s_c2 + 64;                                          // [64] ... with that row or a lower one
  uint32_t *s_slots = s_cum + 64;                                       // [64] sampled slots of the unit (by row)
  uint2 *s_desc = (uint2 *)(s_slots + 64);                              // [R2G_DCAP] descriptor lanes of the partition being scanned
  uint64_t *s_kl = (uint64_t *)(s_desc + R2G_DCAP);                     // [kcap + 2] kept keys
  uint8_t *s_len = (uint8_t *)(s_kl + kcap + 2);                        // [np * 64] sub-row length of (partition, row lane)
  const uint32_t units = bv.nq * bv.nstrand;
  const uint32_t ns_max = prm.ns_max;
  const uint32_t *postings = db.postings;
  unsigned long long n_done_local = 0;
  R2_CLK(unsigned long long tc_pre = 0, tc_scan = 0, tc_sel = 0;)
  R2_CLK2(unsigned long long tq[5] = {0, 0, 0, 0, 0};)

  for (uint32_t k = lane; k < R2G_HB_BITS / 32; k += 64) s_hba[k] = 0;

  uint32_t ubase = 0, uidx = 4;
  for (;;) {
    if (uidx == 4) {
      uint32_t v = 0;
      if (lane == 0) v = (uint32_t)atomicAdd(&bv.counters[UGS_CTR_NEXT_RANK2], 4ull);
      ubase = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
      uidx = 0;
    }
    const uint32_t unit = ubase + uidx;
    ++uidx;
    if (unit >= units) break;
    R2_CLK(const unsigned long long tg0 = clock64(); unsigned long long tg1 = tg0, tg2 = tg0;)
    const uint32_t ns = (uint32_t)__builtin_amdgcn_readfirstlane((int)bv.unit_ns[unit]);
    bool bad = ns > R2G_MAXROWS;
    if (ns == 0) { if (lane == 0) bv.cand_n[unit] = 0; continue; }
    uint32_t nk = 0, n_stg = 0;
    bool any_posting = false;
    if (!bad) {
      // ---- row lanes: lane l < ns owns row ns - 1 - l (DESCENDING rows over ascending lanes)
      const bool rowlane = lane < ns;
      const uint32_t myrow = rowlane ? ns - 1u - lane : 0u;
      const uint32_t slot = rowlane ? bv.unit_slots[(uint64_t)unit * ns_max + myrow] : 0u;
      const uint64_t rs = rowlane ? db.row_off[slot] : 0ull;
      const uint32_t rsb = (uint32_t)(rs * 4u);                            // byte offset of the row in the postings array (< 4 GiB: host check)
      s_slots[lane < ns ? myrow : lane] = slot;                             // (by row: the fill walks rows 0, 1, ...)
      s_c2[lane] = 0; s_cum[lane] = 0;
      // the row's line of the partition table -> sub-row lengths of all partitions in LDS (one byte each).  Eight 16-byte loads in flight
      // together (all lanes load: the others read slot 0's line); words past the line's end (index np) are read at np and never used
      {
        const uint32_t *pp = db.part2 + (uint64_t)slot * (np + 1u);
        uint32_t prev = 0; bool big = false;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if ((uint32_t)h * 32u > np) break;
          u32x4 e[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) { const uint32_t w = (uint32_t)(h * 8 + i) * 4u; __builtin_memcpy(&e[i], pp + (w <= np ? w : np), 16); }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const uint32_t p0 = (uint32_t)(h * 8 + i) * 4u;
            const uint32_t l3 = e[i].x - prev, l0 = e[i].y - e[i].x, l1 = e[i].z - e[i].y, l2 = e[i].w - e[i].z;
            if (p0 > 0u && p0 <= np) { big = big || (rowlane && l3 > 255u); s_len[(p0 - 1u) * 64u + lane] = rowlane ? (uint8_t)l3 : 0; }
            if (p0 < np) { big = big || (rowlane && l0 > 255u); s_len[p0 * 64u + lane] = rowlane ? (uint8_t)l0 : 0; }
            if (p0 + 1u < np) { big = big || (rowlane && l1 > 255u); s_len[(p0 + 1u) * 64u + lane] = rowlane ? (uint8_t)l1 : 0; }
            if (p0 + 2u < np) { big = big || (rowlane && l2 > 255u); s_len[(p0 + 2u) * 64u + lane] = rowlane ? (uint8_t)l2 : 0; }
            prev = e[i].w;
          }
        }
        if (r2_ballot(big)) bad = true;                                      // (a sub-row of more than 255 postings: not a sparse row)
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

      // ---- the chunk cursor: partition cp, chunk cq of it; per row lane the layout of the partition's descriptor lanes
      uint32_t cp = 0xffffffffu, cq = 0, ctot = 0;
      uint32_t cur_off = 0;                                                // row lanes: postings of the row before partition cp
      bool exhausted = false;
      // next chunk: per-lane descriptor (byte offset of the lane's four postings, valid count, row) + the chunk's partition
      auto next_chunk = [&](uint32_t &voff, uint32_t &lmeta, uint32_t &part) -> bool {
        if (exhausted) { voff = 0; lmeta = 0; part = 0xffffffffu; return false; }
        if (cp != 0xffffffffu && (cq + 1u) * 64u < ctot) ++cq;
        else {
          uint32_t r_len, r_nl, r_st;
          for (;;) {
            ++cp;
            if (cp >= np) { exhausted = true; voff = 0; lmeta = 0; part = 0xffffffffu; return false; }
            r_len = rowlane ? (uint32_t)s_len[cp * 64u + lane] : 0u;
            r_nl = (r_len + 3u) >> 2;
            const uint32_t incl = r2_wave_incl_sum(r_nl, lane);
            r_st = incl - r_nl;
            ctot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (ctot) break;
          }
          if (ctot > R2G_DCAP) { bad = true; exhausted = true; voff = 0; lmeta = 0; part = 0xffffffffu; return false; }
          // the partition's descriptor lanes: every row lane writes the quads of its sub-row (byte offset of the quad, valid postings | row << 8)
          const uint32_t base = rsb + cur_off * 4u;
          cur_off += r_len;
          for (uint32_t q = 0; r2_ballot(q < r_nl) != 0ull; ++q) {
            const uint32_t rem = r_len - q * 4u;
            if (q < r_nl) s_desc[r_st + q] = make_uint2(base + q * 16u, (rem < 4u ? rem : 4u) | (myrow << 8));
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          cq = 0;
        }
        const uint32_t d = cq * 64u + lane;
        uint2 e = make_uint2(0u, 0u);
        if (d < ctot) e = s_desc[d];
        voff = e.x; lmeta = e.y; part = cp;
        return true;
      };
      auto zero_bitmap = [&]() {
        uint4 z; z.x = z.y = z.z = z.w = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        unsigned char *zp = smem + lane * 16u;                               // (bm_bytes is a multiple of 1024: straight-line stores)
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) if (k * 1024u < bm_bytes) *(uint4 *)(zp + k * 1024u) = z;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      };
      // ---- a partition is done: group its records by target, make keys, prune, keep (as k_rank2, 64-bit keys)
      auto finalize = [&]() {
        const uint32_t n = n_stg;
        n_stg = 0;
        if (n == 0) return;
        if (n > R2_SCAP) { bad = true; return; }
        if (n == 1u) {
          // the common partition: one record = one count-2 target (no grouping; the running totals are bumped in place)
          const uint32_t rec1 = s_stg[0], row1 = (rec1 >> 24) & 63u;
          if (s_cum[row1] < K) {
            if (nk < kcap) {
              if (lane == 0) s_kl[nk] = ((uint64_t)253u << 32) | rec1;
              if (lane == row1) s_c2[lane] += 1u;
              if (lane >= row1) s_cum[lane] += 1u;
              ++nk;
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            } else bad = true;
          }
          return;
        }
        const bool two = n > 64u;                                          // (a second register batch of records: rare)
        uint32_t rec[2], t[2], row[2], wofs[2], hbit[2], cnt[2], cumv[2]; bool act[2], fl[2], drop[2], shared[2] = {false, false};
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          act[b] = false; rec[b] = 0; t[b] = 0; row[b] = 0; wofs[b] = 0; hbit[b] = 0; cnt[b] = 2; drop[b] = false; cumv[b] = 0; fl[b] = false;
          if (b == 1 && !two) continue;
          const uint32_t i = (uint32_t)b * 64u + lane;
          act[b] = i < n;
          rec[b] = act[b] ? s_stg[i] : 0u;
          t[b] = rec[b] & 0xffffffu; row[b] = rec[b] >> 24;
          const uint32_t h = (t[b] ^ (t[b] >> 11)) & (R2G_HB_BITS - 1u);
          wofs[b] = h >> 5; hbit[b] = 1u << (h & 31u);
          cumv[b] = act[b] ? s_cum[row[b] & 63u] : 0u;
        }
        uint32_t olda[2] = {0, 0};
#pragma unroll
        for (int b = 0; b < 2; ++b) { if (b == 1 && !two) continue; olda[b] = act[b] ? atomicOr(&s_hba[wofs[b]], hbit[b]) : 0u; }
        // (one pass over the filter: "a record with this hash came before me" flags every record but the first of a bucket, and a flagged
        // record's target is compared with ALL records below - the first of the bucket included)
#pragma unroll
        for (int b = 0; b < 2; ++b) { if (b == 1 && !two) continue; fl[b] = act[b] && (olda[b] & hbit[b]) != 0u; }
#pragma unroll
        for (int b = 0; b < 2; ++b) { if (b == 1 && !two) continue; if (act[b]) s_hba[wofs[b]] = 0; }
        // a target's records: count = records + 1, first row = the lowest row a record carries; the representative is the FIRST
        // record (lowest index) among those with that lowest row (several records of a chunk may carry the same row)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          if (bb == 1 && !two) continue;
          uint64_t m = r2_ballot(fl[bb]);
          while (m) {
            const int L = __ffsll((long long)m) - 1;
            m &= m - 1;
            const uint32_t tL = (uint32_t)__builtin_amdgcn_readlane((int)t[bb], L), rL = (uint32_t)__builtin_amdgcn_readlane((int)row[bb], L);
            const uint32_t iL = (uint32_t)bb * 64u + (uint32_t)L;
            uint32_t nsame = 0, better = 0;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              if (b == 1 && !two) continue;
              const bool sb = act[b] && t[b] == tL;
              const uint32_t ib = (uint32_t)b * 64u + lane;
              nsame += (uint32_t)__popcll(r2_ballot(sb));
              better += (uint32_t)__popcll(r2_ballot(sb && (row[b] < rL || (row[b] == rL && ib < iL))));
              if (sb) shared[b] = true;
            }
            if (nsame >= 2u && lane == (uint32_t)L) { cnt[bb] = nsame + 1u; drop[bb] = better != 0u; }
          }
        }
        // the records that were not flagged themselves (the first of their bucket) but share a flagged record's target: same rule,
        // seen from their side - a lane whose record has company takes count and rank from ballots over its own target
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          if (bb == 1 && !two) continue;
          uint64_t m = r2_ballot(act[bb] && !fl[bb] && shared[bb]);
          while (m) {
            const int L = __ffsll((long long)m) - 1;
            m &= m - 1;
            const uint32_t tL = (uint32_t)__builtin_amdgcn_readlane((int)t[bb], L), rL = (uint32_t)__builtin_amdgcn_readlane((int)row[bb], L);
            const uint32_t iL = (uint32_t)bb * 64u + (uint32_t)L;
            uint32_t nsame = 0, better = 0;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              if (b == 1 && !two) continue;
              const bool sb = act[b] && t[b] == tL;
              const uint32_t ib = (uint32_t)b * 64u + lane;
              nsame += (uint32_t)__popcll(r2_ballot(sb));
              better += (uint32_t)__popcll(r2_ballot(sb && (row[b] < rL || (row[b] == rL && ib < iL))));
            }
            if (nsame >= 2u && lane == (uint32_t)L) { cnt[bb] = nsame + 1u; drop[bb] = better != 0u; }
          }
        }
        uint32_t nk_new = nk;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if (b == 1 && !two) continue;
          const uint64_t key = ((uint64_t)(255u - cnt[b]) << 32) | rec[b];
          const bool keep = act[b] && !drop[b] && (cnt[b] >= 3u || cumv[b] < K);
          const uint64_t m = r2_ballot(keep);
          uint32_t pos = nk_new + r2_mbcnt(m);
          pos = pos < kcap ? pos : kcap;
          if (keep) s_kl[pos] = key;
          if (keep && cnt[b] == 2u) atomicAdd(&s_c2[row[b] & 63u], 1u);
          nk_new += (uint32_t)__popcll(m);
        }
        nk = nk_new;
        if (nk > kcap) { bad = true; return; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        s_cum[lane] = r2_wave_incl_sum(s_c2[lane], lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      };

      // ---- the ring (four accumulator-register slots as k_rank2); per slot the lanes' descriptors and the chunk's partition
      uint32_t lm[4] = {0, 0, 0, 0}, pt[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
      { uint32_t vo;
        next_chunk(vo, lm[0], pt[0]); r2_issue<0>(vo, postings); next_chunk(vo, lm[1], pt[1]); r2_issue<1>(vo, postings);
        next_chunk(vo, lm[2], pt[2]); r2_issue<2>(vo, postings); next_chunk(vo, lm[3], pt[3]); r2_issue<3>(vo, postings); }
      any_posting = pt[0] != 0xffffffffu;
      R2_CLK(tg1 = clock64();)
      uint32_t S_old[4] = {0, 0, 0, 0}, S_bit[4] = {0, 0, 0, 0}, S_t[4] = {0, 0, 0, 0}, S_lm = 0;
      bool pv = false;
      uint32_t cur_part = 0xffffffffu;
      bool scanning = any_posting;
      // records of the counted chunk: a posting that found its bit set; the record carries the LOWEST row among the chunk's postings of
      // its target (the last matching lane: lanes are in descending row order) - the first touch may be any of them
      auto emit_chunk = [&]() {
        const uint32_t myrow8 = S_lm >> 8;
        if (r2_ballot(((S_old[0] & S_bit[0]) | (S_old[1] & S_bit[1]) | (S_old[2] & S_bit[2]) | (S_old[3] & S_bit[3])) != 0u) == 0ull) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool hit = (S_old[j] & S_bit[j]) != 0u;
          uint64_t m = r2_ballot(hit);
          uint32_t rrow = myrow8;
          uint64_t rest = m;
          while (rest) {                                                   // (a handful of hits per unit)
            const int L = __ffsll((long long)rest) - 1;
            rest &= rest - 1;
            const uint32_t tL = (uint32_t)__builtin_amdgcn_readlane((int)S_t[j], L);
            uint64_t mm = 0;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) mm |= r2_ballot(S_bit[jj] != 0u && S_t[jj] == tL);
            const int last = 63 - __builtin_clzll(mm);                      // (mm holds lane L itself)
            const uint32_t low = (uint32_t)__builtin_amdgcn_readlane((int)myrow8, last);
            if (lane == (uint32_t)L) rrow = low;
          }
          { const uint32_t pos = n_stg + r2_mbcnt(m); if (hit && pos < R2_SCAP + 64u) s_stg[pos] = S_t[j] | (rrow << 24); }    // (more than R2_SCAP: the unit is deferred in finalize)
          n_stg += (uint32_t)__popcll(m);
        }
      };
#define R2G_STAGE(k)                                                                                             \
      if (scanning) {                                                                                            \
        const uint32_t part = pt[k];                                                                             \
        if (part == 0xffffffffu || bad) scanning = false;                                                        \
        else {                                                                                                   \
          uint32_t T[4];                                                                                         \
          R2_CLK2(const unsigned long long q0 = clock64();)                                                      \
          r2_take<k>(T);                                                                                         \
          R2_CLK2(const unsigned long long q1 = clock64();)                                                      \
          const uint32_t tlm = lm[k];                                                                            \
          { uint32_t vo; next_chunk(vo, lm[k], pt[k]); r2_issue<k>(vo, postings); }                              \
          R2_CLK2(const unsigned long long q2 = clock64();)                                                      \
          if (pv) { if (R2_UG(n_stg) <= R2_SCAP) emit_chunk(); else bad = true; pv = false; }                    \
          R2_CLK2(const unsigned long long q3 = clock64();)                                                      \
          if (part != cur_part) { finalize(); zero_bitmap(); cur_part = part; }                                  \
          R2_CLK2(const unsigned long long q4 = clock64();)                                                      \
          if (!bad) {                                                                                            \
            const uint32_t nvalid = tlm & 0xffu, sub = part * G;                                                 \
            uint32_t ad[4];                                                                                      \
            S_lm = tlm;                                                                                          \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                      \
              const uint32_t t = T[j];                                                                           \
              S_t[j] = t;                                                                                        \
              /* (the quad's tail belongs to the next partition or row: a word of the lane's own, bit 0) */      \
              ad[j] = (uint32_t)j < nvalid ? (((t - sub) >> 3) & ~3u) : lane * 4u;                               \
              S_bit[j] = (uint32_t)j < nvalid ? (1u << (t & 31u)) : 0u;                                          \
            }                                                                                                    \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) S_old[j] = __hip_atomic_fetch_or((lds32)(uintptr_t)ad[j], S_bit[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
            pv = true;                                                                                           \
          }                                                                                                      \
          R2_CLK2(tq[0] += q1 - q0; tq[1] += q2 - q1; tq[2] += q3 - q2; tq[3] += q4 - q3; tq[4] += clock64() - q4;) \
        }                                                                                                        \
      }
#define R2_UG(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
      while (scanning) { R2G_STAGE(0) R2G_STAGE(1) R2G_STAGE(2) R2G_STAGE(3) }
#undef R2G_STAGE
      // every load of the ring has landed before the next unit issues into the same slots
      asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
      if (!bad) {
        if (pv) { if (R2_UG(n_stg) <= R2_SCAP) emit_chunk(); else bad = true; pv = false; }
        if (!bad) finalize();
      }
#undef R2_UG
      R2_CLK(tg2 = clock64();)
    }
    if (bad) {
      for (uint32_t k = lane; k < R2G_HB_BITS / 32; k += 64) s_hba[k] = 0;
      if (lane == 0) {
        const unsigned long long idx = atomicAdd(&bv.counters[UGS_CTR_DEFER], 1ull);
        bv.defer_list[idx] = unit;
      }
      continue;
    }
    ++n_done_local;
    r2g_finish_unit(db, bv, postings, unit, lane, ns, K, nk, any_posting, s_kl, s_fpk, s_sel, s_slots);
    R2_CLK(tc_pre += tg1 - tg0; tc_scan += tg2 - tg1; tc_sel += clock64() - tg2;)
  }
#ifdef R2_CLOCKS
  if (lane == 0) { atomicAdd(&bv.counters[UGS_CTR_T0], tc_pre); atomicAdd(&bv.counters[UGS_CTR_T1], tc_scan); atomicAdd(&bv.counters[UGS_CTR_T3], tc_sel); }
#endif
#ifdef R2_CLOCKS2     // the stage's sections: take (posting wait) | next chunk + issue | emission | partition boundary | count
  if (lane == 0) { atomicAdd(&bv.counters[UGS_CTR_T0], tq[0]); atomicAdd(&bv.counters[UGS_CTR_T1], tq[1]); atomicAdd(&bv.counters[UGS_CTR_T2], tq[2]); atomicAdd(&bv.counters[UGS_CTR_T3], tq[3]); atomicAdd(&bv.counters[UGS_CTR_T4], tq[4]); }
#endif
  if (lane == 0 && n_done_local) atomicAdd(&bv.counters[UGS_CTR_R2_DONE], n_done_local);
}

// post16[i] = postings[i] mod G: a posting's offset inside its partition (what k_rank2<.., P16> streams); same element positions
__global__ void k_post16(const uint32_t *postings, uint64_t n, uint32_t G, uint16_t *out)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (uint16_t)(postings[i] % G);
}

int ugs_build_post16(const uint32_t *d_postings, uint64_t n, uint32_t G, uint16_t *d_out, hipStream_t st)
{
  if (!n) return UGS_OK;
  if (G == 0 || G > 65536u) { ugs_set_error("16-bit postings need a partition size <= 65536"); return UGS_E_ENVELOPE; }
  hipLaunchKernelGGL(k_post16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_postings, n, G, d_out);
  HIPCHK(hipGetLastError());
  return UGS_OK;
}

static const void *rank2_kernel(int gather = 0, int cl = 0, int p16 = 0, int hv = 0)
{
  if (hv) return (const void *)k_rank2<UGS_R2_DEPTH, true, false, true>;
  return gather ? (const void *)k_rank2g : cl ? (const void *)k_rank2<UGS_R2_DEPTH, true, false>
                : p16 ? (const void *)k_rank2<UGS_R2_DEPTH, false, true> : (const void *)k_rank2<UGS_R2_DEPTH, false, false>;
}

size_t ugs_rank2_lds(uint32_t G, uint32_t kcap, uint32_t clcap, int cl)
{
  return (size_t)G / 8 + (size_t)r2_fixed_words(cl != 0) * 4 + (size_t)clcap * 8 + ((size_t)kcap + 4) * 4;       // (the kernel's own carve)
}

size_t ugs_rank2_hv_lds(uint32_t gsize, uint32_t kcap, uint32_t clcap)
{
  return (size_t)gsize / 2 + (size_t)r2_fixed_words(true) * 4 + (size_t)clcap * 8 + ((size_t)kcap + 4) * 4;      // (the kernel's own carve, bm_bytes = gsize / 2)
}

size_t ugs_rank2g_lds(uint32_t G, uint32_t kcap, uint32_t np)
{
  return (size_t)G / 8 + ((size_t)R2_SCAP + 64 + R2G_HB_BITS / 32 + 64 * 3) * 4 + (size_t)R2G_DCAP * 8 + ((size_t)kcap + 2) * 8 + (size_t)np * 64;      // (must mirror the kernel's carve)
}

int ugs_rank2_blocks_per_cu(size_t lds, int gather, int cl, int p16, int hv)
{
  int n = 0;
  const void *fn = rank2_kernel(gather, cl, p16, hv);
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 64, lds) != hipSuccess || n < 1) n = 1;
  return n;
}

int ugs_launch_rank2(const UgsDbView &db, const UgsBatchView &b, const UgsRank2Params &prm, int grid, hipStream_t st)
{
  const bool cl = b.cand_key != nullptr;                               // cluster_fast's walk records (ugs_cluster.cpp)
  if (cl && (prm.gather || !b.cl_ev || !b.cl_info || prm.kcap > 508u)) { ugs_set_error("bitmap ranking kernel: cluster mode outside its envelope"); return UGS_E_ENVELOPE; }
  const bool p16 = prm.post16 != nullptr;
  if (p16 && (cl || prm.gather || prm.np > R2_P16_MAXNP || prm.G > 65536u)) { ugs_set_error("bitmap ranking kernel: 16-bit postings outside their envelope"); return UGS_E_ENVELOPE; }
  const void *fn = rank2_kernel((int)prm.gather, cl ? 1 : 0, p16 ? 1 : 0);
  const size_t need = prm.gather ? ugs_rank2g_lds(prm.G, prm.kcap, prm.np) : ugs_rank2_lds(prm.G, prm.kcap, prm.clcap, cl ? 1 : 0);
  if (prm.lds < need) { ugs_set_error("bitmap ranking kernel: %u bytes of LDS per wave, its carve needs %zu", prm.lds, need); return UGS_E_ENVELOPE; }
  HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)prm.lds));
  UgsDbView a0 = db; UgsBatchView a1 = b; UgsRank2Params a2 = prm;
  void *args[] = {&a0, &a1, &a2};
  if (ugs_kernel_log) ugs_before_launch("k_rank2 / k_rank2g");
  HIPCHK(hipLaunchKernel(fn, dim3(grid), dim3(64), args, prm.lds, st));
  if (ugs_kernel_log) ugs_after_launch("k_rank2 / k_rank2g", st);
  HIPCHK(hipGetLastError());
  if (cl && prm.hv_grid > 0 && prm.defer2) {
    // the heavy units of cluster_fast: the deferred list of the kernel above goes through the counting instantiation; what that cannot take
    // (defer2) is k_rank's (ugs_launch_rank: use_defer = 2)
    if (db.np > R2_HV_MAXNP || (db.gsize & 63u) || db.gsize > 65536u) { ugs_set_error("heavy-unit ranking kernel: partition table outside its envelope"); return UGS_E_ENVELOPE; }
    const size_t need_hv = ugs_rank2_hv_lds(db.gsize, prm.kcap, prm.clcap);
    if (prm.hv_lds < need_hv) { ugs_set_error("heavy-unit ranking kernel: %u bytes of LDS per wave, its carve needs %zu", prm.hv_lds, need_hv); return UGS_E_ENVELOPE; }
    const void *fh = rank2_kernel(0, 1, 0, 1);
    HIPCHK(hipFuncSetAttribute(fh, hipFuncAttributeMaxDynamicSharedMemorySize, (int)prm.hv_lds));
    if (ugs_kernel_log) ugs_before_launch("k_rank2<HV>");
    HIPCHK(hipLaunchKernel(fh, dim3(prm.hv_grid), dim3(64), args, prm.hv_lds, st));
    if (ugs_kernel_log) ugs_after_launch("k_rank2<HV>", st);
    HIPCHK(hipGetLastError());
  }
  return UGS_OK;
}
