// ugs_rank.hip - U-sort candidate ranking on gfx950: query words -> posting scan ->
// per-target word counts -> the first K candidates in the reference's exact order.
//
// Replaces (reference, /root/reference/src):
//   UDBSearcher::SetQueryWordsAllNoBadNoPattern / SetQueryUniqueWords   udbsearcher.cpp:128-194
//   GetWordCountingParams                                              wordparams.cpp:167-192 (host table)
//   UDBUsortedSearcher::UDBSearchBig  (scan + CountSortSubsetDesc)     udbusortedsearcherbig.cpp:31-110
//   UDBUsortedSearcher::SetU_NonCoded / SetTopBump / CountSortOrderDesc udbusortedsearcher.cpp:230-267,375-410
//   CountSortSubsetDesc / CountSortOrderDesc                           countsort.cpp:6-191
//
// Design (DESIGN.md "K2/K3"): one workgroup per (query, strand).  The target space is cut into
// partitions of gsize targets; each WAVE owns a private LDS counter table (4/8/16-bit
// counters, width chosen per query) for one partition at a time and handles the sampled rows'
// sub-rows (cut out by the index's partition table, no searching):
//   pass 1  all sub-row loads of the partition are issued back-to-back (one register per row,
//           one posting per lane), then one LDS atomic increment per posting (the reference's U[t]++)
//   pass 2  read the final count of every posting, clear the counters (the table is clean for the
//           next partition - no bulk zeroing), record the first-touch position per count value and
//           emit every posting whose target has count >= 2
// Nothing in the scan needs a workgroup barrier or a row order: a target with count c is emitted c
// times and the duplicates are dropped during selection (the smallest key of a target is its
// first touch).  The reference fully sorts ~38 k touched targets per query but its candidate loop
// can consume at most K = maxaccepts+maxrejects-1 of them, so only the K smallest keys
// (count desc, first-touch position asc) are selected, after applying the reference's
// "MinValue = prevMax/2" (and, on the small path, -bump) cut-offs exactly.
#include <atomic>
#include "ugs_dev.h"
#include "ugs_rank2.h"
#include <cstdlib>
#include <cstdio>
#include <algorithm>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ugs_set_error("%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return UGS_E_HIP; } } while (0)
#define RCCHK_(x) do { int rc_ = (x); if (rc_ != UGS_OK) return rc_; } while (0)

#include "ugs_rank_keys.h"
#ifndef UGS_RANK_TU
#define UGS_RANK_TU 0       // see the end of this file
#endif
#define KEY_INF 0xffffffffffffffffull
#define RB 16                  // rows per register batch
#define SEL_REGS 8             // emitted entries held per thread during block-wide selection
#define SELQ 4                 // entries per lane per wave in the register-resident selection
#define TINV 0xffffffffu

template <int BITS> struct Tbl {
  static constexpr uint32_t MASK = (1u << BITS) - 1u;
  static __device__ __forceinline__ void inc(uint32_t *t, uint32_t x) {
    atomicAdd(&t[(x * BITS) >> 5], 1u << ((x * BITS) & 31));
  }
  static __device__ __forceinline__ uint32_t get(const uint32_t *t, uint32_t x) {
    return (t[(x * BITS) >> 5] >> ((x * BITS) & 31)) & MASK;
  }
  static __device__ __forceinline__ void clear(uint32_t *t, uint32_t x) {
    atomicAnd(&t[(x * BITS) >> 5], ~(MASK << ((x * BITS) & 31)));
  }
};

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src)
{
  uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, src);
  uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}

// wave-wide inclusive prefix sum with DPP row shifts (no LDS crossbar round trips)
__device__ __forceinline__ uint32_t wave_incl_sum_u32(uint32_t v)
{
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
  // add the totals of the preceding 16-lane rows
  const uint32_t t0 = __builtin_amdgcn_readlane((int)v, 15), t1 = __builtin_amdgcn_readlane((int)v, 31), t2 = __builtin_amdgcn_readlane((int)v, 47);
  const uint32_t row = (uint32_t)(threadIdx.x & 63) >> 4;
  return v + (row >= 1 ? t0 : 0u) + (row >= 2 ? t1 : 0u) + (row >= 3 ? t2 : 0u);
}

struct RankShared {
  uint32_t emit_n, M, next_value, min_value, n_sel, nev, exhausted, pad1;
  uint64_t fill_limit;       // small path: count-1 targets kept only below this position
  uint64_t last_key;
  uint64_t red[8];
  uint32_t wsum[8];
  uint32_t ncl, qcut, pad2, pad3;
  uint32_t wn[8];            // emitted entries per wave (each wave fills its own segment of the candidate buffer)
  uint32_t hist[256];        // emitted entries per (count, first row) class: hist[c*16 + i], 4-bit fast path only
};

struct ScanCtx {
  const PairQ *pq;           // non-null: small path with pair filters
  const uint64_t *row_off; const uint32_t *part; const uint32_t *postings;
  const uint32_t *s_slots; const uint32_t *s_part; uint32_t *tbl; unsigned long long *s_fp; RankShared *sh;
  uint32_t *hist;            // non-null: count emitted entries per (count,row) class
  uint64_t *ebuf; uint64_t ecap;
  uint64_t *s_ebuf;          // this wave's LDS segment: its first `elw` emitted keys of a unit (the usual case), the rest go to ebuf
  uint32_t *wn;              // this wave's emitted-entry count (a register of the kernel: slots are handed out without an atomic)
  uint32_t elw; uint64_t ecapw;   // entries of the LDS segment / of this wave's share of the HBM buffer
  uint32_t ns, np, gsize, tbl_words;
  int wave, wpb, lane; bool small_path;
};

#ifndef UGS_ELDS
#define UGS_ELDS 1024u
#endif
#ifndef UGS_ELDS_HOT
#define UGS_ELDS_HOT 128u      // keys of a unit that stay in LDS in the HOT instantiation (its LDS must fit six workgroups per CU)
#endif
__device__ __forceinline__ void put_key(const ScanCtx &s, uint64_t idx, uint64_t key)     // idx: position in this wave's segment
{
  if (idx < s.elw) s.s_ebuf[idx] = key;
  else if (idx - s.elw < s.ecapw) s.ebuf[idx - s.elw] = key;
}

// emit the lanes with e==true (at most `take` of them, in lane order) into the WG's candidate buffer
__device__ __forceinline__ void emit_lanes(const ScanCtx &s, bool e, uint32_t take_cap, uint64_t key)
{
  const uint64_t m = __ballot(e);
  if (!m) return;
  const uint32_t n = __popcll(m);
  const uint32_t take = n < take_cap ? n : take_cap;
  if (!take) return;
  const uint32_t rank = __popcll(m & ((1ull << s.lane) - 1ull));
  const uint32_t base = *s.wn;
  *s.wn = base + take;
  if (e && rank < take) put_key(s, (uint64_t)base + rank, key);
}

// sub-row [a,b) of row `slot` restricted to targets [lo_t, hi_t); the partition table gives the
// partition bounds, a binary search narrows them only when the LDS table is smaller than a partition
__device__ __forceinline__ void row_bounds(const ScanCtx &s, uint32_t slot, uint32_t p, bool split,
                                           uint32_t lo_t, uint32_t hi_t, uint64_t &a, uint64_t &b)
{
  const uint64_t rb = s.row_off[slot];
  const uint32_t *pp = s.part + (uint64_t)slot * (s.np + 1) + p;
  a = rb + pp[0]; b = rb + pp[1];
  if (split) {
    uint64_t l = a, h = b;
    while (l < h) { uint64_t m = (l + h) >> 1; if (s.postings[m] < lo_t) l = m + 1; else h = m; }
    const uint64_t na = l;
    h = b;
    while (l < h) { uint64_t m = (l + h) >> 1; if (s.postings[m] < hi_t) l = m + 1; else h = m; }
    a = na; b = l;
  }
}

// pass-2 work for one posting per lane (t valid iff `on`)
template <int CB, bool FILL>
__device__ __forceinline__ void extract_one(const ScanCtx &s, bool on, uint32_t t, uint32_t base_t, uint32_t i,
                                            uint32_t c, uint32_t &quota_left, uint64_t fill_limit)
{
  const uint64_t pos = s.small_path ? (uint64_t)t : (((uint64_t)i << 32) | t);
  if (!FILL) {
    // (first positions and class sizes of the count >= 2 entries are derived from the emitted keys after the scan)
    if (on && c == 1) { if (pos < s.s_fp[1]) atomicMin(&s.s_fp[1], (unsigned long long)pos); }
    emit_lanes(s, on && c >= 2, 0xffffffffu, make_key(c, pos));
  } else {
    const bool e = on && c == 1 && pos < fill_limit && !(s.pq && pair_reject(*s.pq, t));
    const uint32_t n = __popcll(__ballot(e));
    emit_lanes(s, e, quota_left, make_key(1, pos));
    quota_left = quota_left > n ? quota_left - n : 0;
  }
}

// Row-major flattening of up to 64 sub-rows [a_r, b_r) held one per lane: posting f of the concatenation belongs to
// the row r with excl_r <= f < incl_r (binary search over the lanes' inclusive prefix sums with ds_bpermute, no memory).
// Used when the sub-rows are short (protein indexes: 1-6 postings per row and range), where a loop over rows would
// spend an instruction stream per row to move a handful of postings.
struct FlatRows { uint32_t incl, excl, T; uint64_t a; };
__device__ __forceinline__ bool flat_rows_setup(FlatRows &fr, uint64_t a, uint64_t b, uint64_t rows)
{
  const uint32_t len = (uint32_t)(b - a);
  fr.incl = wave_incl_sum_u32(len); fr.excl = fr.incl - len; fr.a = a;
  fr.T = (uint32_t)__builtin_amdgcn_readlane((int)fr.incl, 63);
  return rows != 0 && fr.T < 32u * (uint32_t)__popcll(rows);      // average sub-row below half a wavefront
}
__device__ __forceinline__ bool flat_rows_at(const FlatRows &fr, uint32_t f, uint32_t &row, uint64_t &k)
{
  uint32_t lo = 0, hi = 63;
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const uint32_t mid = (lo + hi) >> 1;
    const uint32_t v = (uint32_t)__shfl((int)fr.incl, (int)mid);
    if (v > f) hi = mid; else lo = mid + 1;
  }
  row = lo;
  const uint32_t ex = (uint32_t)__shfl((int)fr.excl, (int)lo);
  const uint32_t alo = (uint32_t)__shfl((int)(uint32_t)fr.a, (int)lo), ahi = (uint32_t)__shfl((int)(uint32_t)(fr.a >> 32), (int)lo);
  k = (((uint64_t)ahi << 32) | alo) + (f - ex);
  return f < fr.T;
}

// Generic handling of one table range [base_t, hi_t) of partition p: any number of rows, any
// sub-row length; rows are walked in order and counters are cleared right after they are read, so
// only the first touch of a target sees a non-zero count (no duplicates are emitted).
// FILL=true: emit count-1 targets in scan order, at most `need` of them.
// FLATPF: the sparse-dictionary kernel (protein) requests the flattened postings ahead of use
template <int CB, bool FILL, bool BATCH = false, bool FLATPF = false>
__device__ __forceinline__ void range_generic(const ScanCtx &s, uint32_t p, bool split, uint32_t base_t, uint32_t hi_t,
                                              uint32_t need, uint64_t fill_limit, bool have_ab = false, uint64_t a_in = 0,
                                              uint64_t b_in = 0)
{
  constexpr uint32_t EPW = 32 / CB;
  constexpr int NB = 4;                                        // rows whose first loads are issued together (BATCH)
  const int lane = s.lane;
  const uint32_t ns = s.ns;
  uint32_t *tbl = s.tbl;
  const uint32_t *postings = s.postings;
  uint32_t quota_left = need;
  if constexpr (FLATPF && !FILL) {
    // sparse rows, at most 64 of them, and the whole range holds at most two instructions' worth of postings (the usual case
    // of a protein index): bounds, flattening and postings are fetched ONCE and serve the count pass and the extract pass
    if (ns <= 64) {
      uint64_t a = 0, b = 0;
      if (have_ab) { a = a_in; b = b_in; }
      else if ((uint32_t)lane < ns) row_bounds(s, s.s_slots[lane], p, split, base_t, hi_t, a, b);
      const uint64_t rows = __ballot(b > a);
      if (!rows) return;
      FlatRows fr;
      if (flat_rows_setup(fr, a, b, rows) && fr.T <= 128) {
        uint32_t row[2] = {0, 0}, t[2] = {0, 0}; bool on[2]; uint64_t k0, k1;
        on[0] = flat_rows_at(fr, (uint32_t)lane, row[0], k0);
        on[1] = fr.T > 64 && flat_rows_at(fr, 64 + (uint32_t)lane, row[1], k1);
        t[0] = on[0] ? postings[k0] : 0u; t[1] = on[1] ? postings[k1] : 0u;
        if (on[0]) Tbl<CB>::inc(tbl, t[0] - base_t);
        if (on[1]) Tbl<CB>::inc(tbl, t[1] - base_t);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (h == 1 && fr.T <= 64) break;
          const bool o2 = on[h];
          const uint32_t t2 = t[h];
          const uint32_t c2 = o2 ? Tbl<CB>::get(tbl, t2 - base_t) : 0u;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          if (o2 && c2) Tbl<CB>::clear(tbl, t2 - base_t);
          // an instruction spans several rows: a target held by two of its lanes (count >= 2) is reported by the lower lane = the
          // earlier row only, as the row-by-row walk would
          bool dup = false;
          uint64_t m = __ballot(o2 && c2 >= 2);
          while (m) {
            const int L = __ffsll((long long)m) - 1;
            const uint32_t tL = (uint32_t)__builtin_amdgcn_readlane((int)t2, L);
            const bool same = o2 && lane > L && t2 == tL;
            dup = dup || same;
            m &= ~(__ballot(same) | (1ull << L));
          }
          extract_one<CB, FILL>(s, o2 && !dup, t2, base_t, row[h], c2, quota_left, fill_limit);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        return;
      }
    }
  }
  for (uint32_t i0 = 0; i0 < ns; i0 += 64) {
    uint64_t a = 0, b = 0;
    if (have_ab) { a = a_in; b = b_in; }                       // (ns <= 64: the caller tracks each row's cursor)
    else if (i0 + lane < ns) row_bounds(s, s.s_slots[i0 + lane], p, split, base_t, hi_t, a, b);
    uint64_t rows = __ballot(b > a);                           // only rows with postings in this range, in row order
    FlatRows fr;
    if (flat_rows_setup(fr, a, b, rows)) {
      // sparse sub-rows: lanes = postings of ALL rows of this range in row-major order
      if constexpr (FLATPF) {
        // two instructions' worth of postings are located and requested before either is counted
        for (uint32_t f0 = 0; f0 < fr.T; f0 += 128) {
          uint32_t row0, row1; uint64_t k0, k1;
          const bool on0 = flat_rows_at(fr, f0 + (uint32_t)lane, row0, k0);
          const bool on1 = f0 + 64 < fr.T && flat_rows_at(fr, f0 + 64 + (uint32_t)lane, row1, k1);
          const uint32_t t0 = on0 ? postings[k0] : 0u, t1 = on1 ? postings[k1] : 0u;
          if (on0) Tbl<CB>::inc(tbl, t0 - base_t);
          if (on1) Tbl<CB>::inc(tbl, t1 - base_t);
        }
        continue;
      }
      for (uint32_t f0 = 0; f0 < fr.T; f0 += 64) {
        uint32_t row; uint64_t k;
        const bool on = flat_rows_at(fr, f0 + (uint32_t)lane, row, k);
        if (on) Tbl<CB>::inc(tbl, postings[k] - base_t);
      }
      continue;
    }
    if constexpr (!BATCH) {                                    // (launches with 4-bit tables keep the shape the hot path was tuned with)
      while (rows) {
        const uint32_t r = (uint32_t)__ffsll((long long)rows) - 1;
        rows &= rows - 1;
        const uint64_t ra = shfl64(a, r), rbb = shfl64(b, r);
        for (uint64_t k = ra + lane; k < rbb; k += 64) Tbl<CB>::inc(tbl, postings[k] - base_t);
      }
    } else {
      // four rows at a time: their first 64 postings are requested together, so a wave keeps four loads in flight
      // instead of waiting out one load per row (the mid-identity configurations, 16-255 sampled rows, live here)
      while (rows) {
        uint64_t ra[NB], rbb[NB]; uint32_t v[NB]; bool on[NB];
  #pragma unroll
        for (int u = 0; u < NB; ++u) {
          ra[u] = 0; rbb[u] = 0;
          if (rows) {
            const uint32_t r = (uint32_t)__ffsll((long long)rows) - 1;
            rows &= rows - 1;
            ra[u] = shfl64(a, r); rbb[u] = shfl64(b, r);
          }
          on[u] = ra[u] + lane < rbb[u];
          v[u] = on[u] ? postings[ra[u] + lane] : 0u;
        }
  #pragma unroll
        for (int u = 0; u < NB; ++u) {
          if (on[u]) Tbl<CB>::inc(tbl, v[u] - base_t);
          for (uint64_t k = ra[u] + 64 + lane; k < rbb[u]; k += 64) Tbl<CB>::inc(tbl, postings[k] - base_t);
        }
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (FILL && s.small_path) {
    // small path scans U[] in ascending target order (udbusortedsearcher.cpp:230-267): walk the
    // table itself, emit the first `need` count-1 targets of this range, clear as we go
    const uint32_t words = s.tbl_words;
    for (uint32_t w0 = 0; w0 < words; w0 += 64) {
      const uint32_t wi = w0 + lane;
      uint32_t word = 0;
      if (wi < words) { word = tbl[wi]; if (word) tbl[wi] = 0; }
      uint32_t mine = 0;
      for (uint32_t e2 = 0; e2 < EPW; ++e2) {
        const uint32_t cc = (word >> (e2 * CB)) & Tbl<CB>::MASK;
        const uint64_t tt = (uint64_t)base_t + (uint64_t)wi * EPW + e2;
        if (cc == 1 && tt < fill_limit && !(s.pq && pair_reject(*s.pq, (uint32_t)tt))) ++mine;
      }
      uint32_t incl = mine;
      for (int o = 1; o < 64; o <<= 1) { uint32_t x = __shfl_up((int)incl, o); if (lane >= o) incl += x; }
      const uint32_t total = __builtin_amdgcn_readlane((int)incl, 63);
      if (total && quota_left) {
        const uint32_t take = quota_left < total ? quota_left : total;
        const uint32_t base = *s.wn;
        *s.wn = base + take;
        uint32_t rr = incl - mine;
        for (uint32_t e2 = 0; e2 < EPW; ++e2) {
          const uint32_t cc = (word >> (e2 * CB)) & Tbl<CB>::MASK;
          const uint64_t tt = (uint64_t)base_t + (uint64_t)wi * EPW + e2;
          if (cc == 1 && tt < fill_limit && !(s.pq && pair_reject(*s.pq, (uint32_t)tt))) {
            if (rr < take) put_key(s, (uint64_t)base + rr, make_key(1, tt));
            ++rr;
          }
        }
      }
      quota_left = quota_left > total ? quota_left - total : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return;
  }
  for (uint32_t i0 = 0; i0 < ns; i0 += 64) {
    uint64_t a = 0, b = 0;
    if (have_ab) { a = a_in; b = b_in; }
    else if (i0 + lane < ns) row_bounds(s, s.s_slots[i0 + lane], p, split, base_t, hi_t, a, b);
    uint64_t rows = __ballot(b > a);
    FlatRows fr;
    if (flat_rows_setup(fr, a, b, rows)) {
      uint32_t nrow = 0, nt2 = 0; bool no2 = false;             // (FLATPF) the next instruction's postings, requested one trip ahead
      if constexpr (FLATPF) { uint64_t k; no2 = flat_rows_at(fr, (uint32_t)lane, nrow, k); nt2 = no2 ? postings[k] : 0u; }
      for (uint32_t f0 = 0; f0 < fr.T; f0 += 64) {
        uint32_t row; uint64_t k;
        bool o2;
        uint32_t t2 = 0, c2 = 0;
        if constexpr (FLATPF) {
          o2 = no2; row = nrow; t2 = nt2;
          no2 = false;
          if (f0 + 64 < fr.T) { no2 = flat_rows_at(fr, f0 + 64 + (uint32_t)lane, nrow, k); nt2 = no2 ? postings[k] : 0u; }
          if (o2) c2 = Tbl<CB>::get(tbl, t2 - base_t);
        } else {
          o2 = flat_rows_at(fr, f0 + (uint32_t)lane, row, k);
          if (o2) { t2 = postings[k]; c2 = Tbl<CB>::get(tbl, t2 - base_t); }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (o2 && c2) Tbl<CB>::clear(tbl, t2 - base_t);
        // one instruction now spans several rows: a target held by two of its lanes (count >= 2) is reported by the
        // lower lane = the earlier row only, as the row-by-row walk would
        bool dup = false;
        uint64_t m = __ballot(o2 && c2 >= 2);
        while (m) {
          const int L = __ffsll((long long)m) - 1;
          const uint32_t tL = (uint32_t)__builtin_amdgcn_readlane((int)t2, L);
          const bool same = o2 && lane > L && t2 == tL;
          dup = dup || same;
          m &= ~(__ballot(same) | (1ull << L));
        }
        extract_one<CB, FILL>(s, o2 && !dup, t2, base_t, i0 + row, c2, quota_left, fill_limit);
      }
      continue;
    }
    if constexpr (!BATCH) {
      while (rows) {
        const uint32_t r = (uint32_t)__ffsll((long long)rows) - 1;
        rows &= rows - 1;
        const uint64_t ra = shfl64(a, r), rbb = shfl64(b, r);
        for (uint64_t k0 = ra; k0 < rbb; k0 += 64) {
          const uint64_t k = k0 + lane;
          const bool o2 = k < rbb;
          uint32_t t2 = 0, c2 = 0;
          if (o2) { t2 = postings[k]; c2 = Tbl<CB>::get(tbl, t2 - base_t); }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          if (o2 && c2) Tbl<CB>::clear(tbl, t2 - base_t);
          extract_one<CB, FILL>(s, o2, t2, base_t, i0 + r, c2, quota_left, fill_limit);
        }
      }
    } else {
      while (rows) {                                             // rows stay in order (first touch reports the row); loads are batched
        uint64_t ra[NB], rbb[NB]; uint32_t v[NB], rr[NB];
  #pragma unroll
        for (int u = 0; u < NB; ++u) {
          ra[u] = 0; rbb[u] = 0; rr[u] = 0;
          if (rows) {
            rr[u] = (uint32_t)__ffsll((long long)rows) - 1;
            rows &= rows - 1;
            ra[u] = shfl64(a, rr[u]); rbb[u] = shfl64(b, rr[u]);
          }
          v[u] = ra[u] + lane < rbb[u] ? postings[ra[u] + lane] : 0u;
        }
  #pragma unroll
        for (int u = 0; u < NB; ++u) {
          for (uint64_t k0 = ra[u]; k0 < rbb[u]; k0 += 64) {
            const uint64_t k = k0 + lane;
            const bool o2 = k < rbb[u];
            uint32_t t2 = 0, c2 = 0;
            if (o2) { t2 = k0 == ra[u] ? v[u] : postings[k]; c2 = Tbl<CB>::get(tbl, t2 - base_t); }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (o2 && c2) Tbl<CB>::clear(tbl, t2 - base_t);
            extract_one<CB, FILL>(s, o2, t2, base_t, i0 + rr[u], c2, quota_left, fill_limit);
          }
        }
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// One whole partition whose sub-rows may be (much) longer than a wavefront - the rows of the words an abundant family of
// database sequences shares (skewed databases: cluster_fast's centroids, amplicon references).  Rows in scan order, 256
// postings per trip (a 16-byte load per lane: 4 consecutive postings) with the next trip's load already in flight; pass 1
// counts, pass 2 reads and clears each counter at once, so only the first touch of a target sees its count.
// (Only the LONG instantiations of the ranking kernel call it: inlined into the register-resident fast path it costs that
// path 3 % through register allocation even when it never runs, and out of line the call ABI costs far more.)
template <int CB>
__device__ __forceinline__ void range_long(const ScanCtx &s, uint32_t p, uint32_t base_t)
{
  typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
  const int lane = s.lane;
  const uint32_t ns = s.ns;
  uint32_t *tbl = s.tbl;
  unsigned long long cache1 = s.s_fp[1];
  for (int pass = 0; pass < 2; ++pass) {
    for (uint32_t i0 = 0; i0 < ns; i0 += 64) {
      uint64_t a = 0, b = 0;
      if (i0 + lane < ns) row_bounds(s, s.s_slots[i0 + lane], p, false, base_t, 0, a, b);
      uint64_t rows = __ballot(b > a);
      while (rows) {
        const uint32_t r = (uint32_t)__ffsll((long long)rows) - 1;
        rows &= rows - 1;
        const uint64_t ra = shfl64(a, r), rbb = shfl64(b, r);
        const uint32_t row = i0 + r;
        const uint32_t *src = s.postings + ra;
        const uint32_t n = (uint32_t)(rbb - ra);
        // (the array is padded: a 16-byte load may run past the row's end, those lanes are masked by index)
        u32x4 nxt;
        __builtin_memcpy(&nxt, src + (uint32_t)lane * 4u, 16);          // 4-byte aligned 16-byte load
        for (uint32_t k0 = 0; k0 < n; k0 += 256) {
          const u32x4 cur = nxt;
          if (k0 + 256 < n) __builtin_memcpy(&nxt, src + k0 + 256u + (uint32_t)lane * 4u, 16);
          const uint32_t i = k0 + (uint32_t)lane * 4u;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool on = i + j < n;
            const uint32_t t = cur[j];
            if (pass == 0) { if (on) Tbl<CB>::inc(tbl, t - base_t); }
            else {
              uint32_t c = 0;
              if (on) {
                const uint32_t x = t - base_t, sh = (x * CB) & 31u;
                c = (atomicAnd(&tbl[(x * CB) >> 5], ~(Tbl<CB>::MASK << sh)) >> sh) & Tbl<CB>::MASK;     // count, cleared at once
              }
              const uint64_t pos = s.small_path ? (uint64_t)t : (((uint64_t)row << 32) | t);
              const bool f1 = c == 1 && pos < cache1;
              if (__ballot(f1)) {
                if (f1) atomicMin(&s.s_fp[1], (unsigned long long)pos);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                cache1 = s.s_fp[1];
              }
              emit_lanes(s, c >= 2, 0xffffffffu, make_key(c, pos));
            }
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_s_waitcnt(0xc07f);            // every add has completed before the clears start
  }
}

template <int CB, bool FILL, bool BATCH = false, bool FLATPF = false>
__device__ __forceinline__ void scan_generic(const ScanCtx &s, uint32_t need, uint64_t fill_limit)
{
  constexpr uint32_t EPW = 32 / CB;
  const uint32_t G = s.gsize;
  const uint32_t tbl_targets = s.tbl_words * EPW;
  const bool split = tbl_targets < G;
  const uint32_t nsub = split ? (G + tbl_targets - 1) / tbl_targets : 1;
  if (split && s.ns <= 64) {
    // Sparse rows (protein indexes: a row holds far fewer postings than there are table ranges): lane r walks
    // row r with a cursor through the consecutive ranges of a partition, so a range's sub-row bounds cost the
    // postings actually crossed instead of two binary searches per (row, range) and pass.
    for (uint32_t p = s.wave; p < s.np; p += s.wpb) {
      uint64_t cur = 0, end = 0;
      uint32_t vcur = 0xffffffffu;
      if ((uint32_t)s.lane < s.ns) {
        const uint32_t slot = s.s_slots[s.lane];
        const uint64_t rb = s.row_off[slot];
        const uint32_t *pp = s.part + (uint64_t)slot * (s.np + 1) + p;
        cur = rb + pp[0]; end = rb + pp[1];
        if (cur < end) vcur = s.postings[cur];
      }
      for (uint32_t sub = 0; sub < nsub; ++sub) {
        const uint32_t base_t = p * s.gsize + sub * tbl_targets;
        const uint32_t pend = (p + 1) * s.gsize;
        const uint32_t hi_t = base_t + tbl_targets < pend ? base_t + tbl_targets : pend;
        const uint64_t a = cur;
        while (cur < end && vcur < hi_t) { ++cur; vcur = cur < end ? s.postings[cur] : 0xffffffffu; }
        if (!__ballot(cur > a)) continue;                        // no sampled row has a posting in this range
        range_generic<CB, FILL, BATCH, FLATPF>(s, p, split, base_t, hi_t, need, fill_limit, true, a, cur);
      }
    }
    return;
  }
  // (not in the dense 8-bit kernels - BATCH without FLATPF: there it is a rare path and goes without)
  if constexpr (FLATPF || !BATCH) if (!split && s.ns <= 64) {
    // at most one row per lane and the table spans the partition: the lane keeps its row's start and reads the sub-row bounds of
    // the partitions ahead while the current one is counted (one global round trip per partition less, twice: both passes)
    uint64_t rb = 0; const uint32_t *pp = s.part;
    const bool rowlane = (uint32_t)s.lane < s.ns;
    if (rowlane) { const uint32_t slot = s.s_slots[s.lane]; rb = s.row_off[slot]; pp = s.part + (uint64_t)slot * (s.np + 1); }
    if constexpr (FLATPF && !FILL) {
      // sparse rows (protein index): a partition's postings - the sub-rows of all rows flattened row-major, usually fewer than
      // 128 - are located and REQUESTED one partition ahead, so that a partition costs no exposed global round trip at all
      // (bounds two ahead, postings one ahead, both passes on registers)
      struct Ahead { uint64_t a, b; FlatRows fr; bool flat, on[2]; uint32_t row[2], t[2]; };
      auto prepare = [&](uint32_t lo, uint32_t hi, Ahead &A) {
        A.a = rb + lo; A.b = rb + hi;
        const uint64_t rows = __ballot(A.b > A.a);
        A.flat = flat_rows_setup(A.fr, A.a, A.b, rows) && A.fr.T <= 128;
        A.on[0] = A.on[1] = false; A.t[0] = A.t[1] = 0; A.row[0] = A.row[1] = 0;
        if (A.flat) {
          uint64_t k0 = 0, k1 = 0;
          A.on[0] = flat_rows_at(A.fr, (uint32_t)s.lane, A.row[0], k0);
          A.on[1] = A.fr.T > 64 && flat_rows_at(A.fr, 64 + (uint32_t)s.lane, A.row[1], k1);
          A.t[0] = A.on[0] ? s.postings[k0] : 0u; A.t[1] = A.on[1] ? s.postings[k1] : 0u;
        }
      };
      uint32_t quota_left = 0;
      const uint32_t p0 = s.wave, p1 = p0 + s.wpb;
      if (p0 >= s.np) return;
      uint32_t lo = 0, hi = 0, lo1 = 0, hi1 = 0;
      if (rowlane) { lo = pp[p0]; hi = pp[p0 + 1]; if (p1 < s.np) { lo1 = pp[p1]; hi1 = pp[p1 + 1]; } }
      Ahead cur, nxt;
      prepare(lo, hi, cur);
      for (uint32_t p = p0; p < s.np; p += s.wpb) {
        const uint32_t pn = p + s.wpb, pnn = pn + s.wpb;
        // everything requested so far has landed (the current partition's postings have been in flight for a whole trip): the
        // requests issued next are then the only ones outstanding and nothing below waits for them
        __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0)
        uint32_t lo2 = 0, hi2 = 0;
        if (rowlane && pnn < s.np) { lo2 = pp[pnn]; hi2 = pp[pnn + 1]; }
        if (pn < s.np) prepare(lo1, hi1, nxt);
        const uint32_t base_t = p * s.gsize;
        if (!cur.flat) {
          if (__ballot(cur.b > cur.a)) range_generic<CB, FILL, BATCH, FLATPF>(s, p, false, base_t, 0, need, fill_limit, true, cur.a, cur.b);
        } else {
          uint32_t *tbl = s.tbl;
          if (cur.on[0]) Tbl<CB>::inc(tbl, cur.t[0] - base_t);
          if (cur.on[1]) Tbl<CB>::inc(tbl, cur.t[1] - base_t);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (h == 1 && cur.fr.T <= 64) break;
            const bool o2 = cur.on[h];
            const uint32_t t2 = cur.t[h];
            const uint32_t c2 = o2 ? Tbl<CB>::get(tbl, t2 - base_t) : 0u;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (o2 && c2) Tbl<CB>::clear(tbl, t2 - base_t);
            // an instruction spans several rows: a target held by two of its lanes (count >= 2) is reported by the lower lane =
            // the earlier row only, as the row-by-row walk would
            bool dup = false;
            uint64_t m = __ballot(o2 && c2 >= 2);
            while (m) {
              const int L = __ffsll((long long)m) - 1;
              const uint32_t tL = (uint32_t)__builtin_amdgcn_readlane((int)t2, L);
              const bool same = o2 && s.lane > L && t2 == tL;
              dup = dup || same;
              m &= ~(__ballot(same) | (1ull << L));
            }
            extract_one<CB, FILL>(s, o2 && !dup, t2, base_t, cur.row[h], c2, quota_left, fill_limit);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        cur = nxt; lo1 = lo2; hi1 = hi2;
      }
      return;
    }
    uint32_t lo = 0, hi = 0;
    if (rowlane && (uint32_t)s.wave < s.np) { lo = pp[s.wave]; hi = pp[s.wave + 1]; }
    for (uint32_t p = s.wave; p < s.np; p += s.wpb) {
      const uint32_t pn = p + s.wpb;
      uint32_t nlo = 0, nhi = 0;
      if (rowlane && pn < s.np) { nlo = pp[pn]; nhi = pp[pn + 1]; }
      range_generic<CB, FILL, BATCH, FLATPF>(s, p, false, p * s.gsize, 0, need, fill_limit, true, rb + lo, rb + hi);
      lo = nlo; hi = nhi;
    }
    return;
  }
  for (uint32_t p = s.wave; p < s.np; p += s.wpb)
    for (uint32_t sub = 0; sub < nsub; ++sub) {
      const uint32_t base_t = p * s.gsize + sub * tbl_targets;
      const uint32_t pend = (p + 1) * s.gsize;
      const uint32_t hi_t = split ? (base_t + tbl_targets < pend ? base_t + tbl_targets : pend) : 0;
      range_generic<CB, FILL, BATCH, FLATPF>(s, p, split, base_t, hi_t, need, fill_limit);
    }
}

// ---- the hot configuration: <= 15 sampled rows (4-bit counters), table covers a whole partition.
// One register per row holds that row's sub-row (one posting per lane); the loads of the NEXT
// partition are in flight while the current one is counted and extracted.
// ---- the hot configuration: <= 15 sampled rows (4-bit counters), table covers a whole partition.
// Row descriptors (slot, row base pointer, sub-row bounds) are wave-uniform and live in SGPRs: the
// partition table is read with scalar loads, a row's load is "scalar base + lane*4" (no per-row
// address VALU), lanes beyond a sub-row simply read the following postings (the array is padded)
// and are masked out later.  One register per row holds the sub-row; the loads of the NEXT
// partition stay in flight while the current one is counted and extracted (ping-pong batches,
// exactly NR loads per batch so the compiler can wait with s_waitcnt vmcnt(NR)).
// read-only index arrays viewed through the constant address space: a wave-uniform address then
// becomes a scalar load (s_load_*) instead of a 64-lane vector load
typedef const uint32_t __attribute__((address_space(4))) *cptr32;
typedef const uint64_t __attribute__((address_space(4))) *cptr64;

template <int NR> struct Batch { uint32_t v[NR]; uint32_t len[NR]; bool tail; };

// slot: the row's index slot; off: byte offset of the row's line in the partition table (slot * (np + 1) * 4 < 2^32, checked on the
// host); base: the row's first posting
template <int NR> struct Rows { uint32_t slot[NR]; uint32_t off[NR]; const uint32_t *base[NR]; };

// HOT instantiation (Big path, 4-bit counters, uniform rows: C2 / C4): the partition-table entries (start, end of the sub-row) of
// all NR rows for one partition as NR scalar loads that share ONE base pointer (table + p * 4) and take the row's line as their
// SGPR offset operand, then one wait - written as one asm block because the compiler never selects the soffset form of s_load: it
// adds a 64-bit offset to the pointer per row and partition, keeps the NR offsets in spilled SGPR pairs and, out of scalar
// registers, moves the NR row pointers into VGPR pairs.  With the block the kernel fits 96 VGPRs without a spill in the partition
// loop, i.e. FIVE workgroups per CU instead of four - and the loop is a latency chain that more waves hide (DESIGN section 4, round 3:
// 62.7 -> 58.9 ms on C2; the block alone at four workgroups is slower, 65.1 ms: one long wait instead of six short ones).
#define UGS_SL(k) "s_load_dwordx2 %" #k ", %[pp], %[o" #k "]\n"
template <int NR> __device__ __forceinline__ void part_pairs(const void *pp, const Rows<NR> &R, uint64_t (&d)[NR]);
template <> __device__ __forceinline__ void part_pairs<8>(const void *pp, const Rows<8> &R, uint64_t (&d)[8])
{
  asm volatile(UGS_SL(0) UGS_SL(1) UGS_SL(2) UGS_SL(3) UGS_SL(4) UGS_SL(5) UGS_SL(6) UGS_SL(7) "s_waitcnt lgkmcnt(0)"
               : "=&s"(d[0]), "=&s"(d[1]), "=&s"(d[2]), "=&s"(d[3]), "=&s"(d[4]), "=&s"(d[5]), "=&s"(d[6]), "=&s"(d[7])
               : [pp] "s"(pp), [o0] "s"(R.off[0]), [o1] "s"(R.off[1]), [o2] "s"(R.off[2]), [o3] "s"(R.off[3]), [o4] "s"(R.off[4]),
                 [o5] "s"(R.off[5]), [o6] "s"(R.off[6]), [o7] "s"(R.off[7]) : "memory");
}
template <> __device__ __forceinline__ void part_pairs<11>(const void *pp, const Rows<11> &R, uint64_t (&d)[11])
{
  asm volatile(UGS_SL(0) UGS_SL(1) UGS_SL(2) UGS_SL(3) UGS_SL(4) UGS_SL(5) UGS_SL(6) UGS_SL(7) UGS_SL(8) UGS_SL(9) UGS_SL(10) "s_waitcnt lgkmcnt(0)"
               : "=&s"(d[0]), "=&s"(d[1]), "=&s"(d[2]), "=&s"(d[3]), "=&s"(d[4]), "=&s"(d[5]), "=&s"(d[6]), "=&s"(d[7]), "=&s"(d[8]),
                 "=&s"(d[9]), "=&s"(d[10])
               : [pp] "s"(pp), [o0] "s"(R.off[0]), [o1] "s"(R.off[1]), [o2] "s"(R.off[2]), [o3] "s"(R.off[3]), [o4] "s"(R.off[4]),
                 [o5] "s"(R.off[5]), [o6] "s"(R.off[6]), [o7] "s"(R.off[7]), [o8] "s"(R.off[8]), [o9] "s"(R.off[9]), [o10] "s"(R.off[10]) : "memory");
}
template <> __device__ __forceinline__ void part_pairs<12>(const void *pp, const Rows<12> &R, uint64_t (&d)[12])
{
  asm volatile(UGS_SL(0) UGS_SL(1) UGS_SL(2) UGS_SL(3) UGS_SL(4) UGS_SL(5) UGS_SL(6) UGS_SL(7) UGS_SL(8) UGS_SL(9) UGS_SL(10) UGS_SL(11) "s_waitcnt lgkmcnt(0)"
               : "=&s"(d[0]), "=&s"(d[1]), "=&s"(d[2]), "=&s"(d[3]), "=&s"(d[4]), "=&s"(d[5]), "=&s"(d[6]), "=&s"(d[7]), "=&s"(d[8]),
                 "=&s"(d[9]), "=&s"(d[10]), "=&s"(d[11])
               : [pp] "s"(pp), [o0] "s"(R.off[0]), [o1] "s"(R.off[1]), [o2] "s"(R.off[2]), [o3] "s"(R.off[3]), [o4] "s"(R.off[4]),
                 [o5] "s"(R.off[5]), [o6] "s"(R.off[6]), [o7] "s"(R.off[7]), [o8] "s"(R.off[8]), [o9] "s"(R.off[9]), [o10] "s"(R.off[10]),
                 [o11] "s"(R.off[11]) : "memory");
}
#undef UGS_SL

template <int NR, bool SL>
__device__ __forceinline__ void issue_batch(const ScanCtx &s, const Rows<NR> &R, uint32_t p, Batch<NR> &B)
{
  const int lane = s.lane;
  uint32_t mx = 0;
  if constexpr (SL) {
    const uint32_t lane4 = (uint32_t)lane * 4u;
    uint64_t d[NR];
    part_pairs<NR>((const char *)s.part + (uint64_t)p * 4u, R, d);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const uint32_t pa = (uint32_t)d[r], pb = (uint32_t)(d[r] >> 32);
      const uint32_t len = (uint32_t)r < s.ns ? pb - pa : 0u;
      B.len[r] = len;
      mx = len > mx ? len : mx;
      // "scalar row pointer + 32-bit byte offset in a VGPR" (global_load ... saddr): pa * 4 + lane * 4 cannot wrap, a row holds
      // fewer than 2^30 postings (host check)
      B.v[r] = *(const uint32_t *)((const char *)R.base[r] + (pa * 4u + lane4));
    }
  } else {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      cptr32 pp = (cptr32)(uintptr_t)s.part + (uint64_t)R.slot[r] * (s.np + 1) + p;     // uniform address -> scalar load
      const uint32_t pa = pp[0], pb = pp[1];
      const uint32_t len = (uint32_t)r < s.ns ? pb - pa : 0u;
      B.len[r] = len;
      mx = len > mx ? len : mx;
      const uint32_t *rowp = R.base[r] + pa;          // wave-uniform pointer (SALU); the load is "scalar base + lane*4"
      B.v[r] = rowp[(uint32_t)lane];
    }
  }
  B.tail = mx > 64;
}

// A partition of the 4-bit path in which some sub-row is longer than a wavefront (skewed databases: the words an abundant family
// shares - cluster_fast's centroids).  The first 64 postings of every row are in registers already (the batch); only the rows
// that are longer fetch their remaining postings, 256 per trip.  Pass 1 counts, pass 2 reads and clears each counter at once in
// ROW ORDER (a row's chunks together), so only the first touch of a target - its lowest row - sees the count.  Replaces a
// walk of ALL rows from memory, twice, which waited out a load latency per row and pass (LONG instantiations only).
template <int NR>
__device__ __forceinline__ void tail_mixed(const ScanCtx &s, const Rows<NR> &R, const Batch<NR> &B, uint32_t p, uint32_t base_t)
{
  typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
  const int lane = s.lane;
  uint32_t *tbl = s.tbl;
  unsigned long long cache1 = s.s_fp[1];
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const uint32_t len = B.len[r];
      if (len == 0) continue;
      auto one = [&](bool on, uint32_t t) {
        if (pass == 0) { if (on) Tbl<4>::inc(tbl, t - base_t); }
        else {
          uint32_t c = 0;
          if (on) { const uint32_t x = t - base_t, sh = (x * 4u) & 31u; c = (atomicAnd(&tbl[(x * 4u) >> 5], ~(15u << sh)) >> sh) & 15u; }
          const uint64_t pos = s.small_path ? (uint64_t)t : (((uint64_t)r << 32) | t);
          const bool f1 = c == 1 && pos < cache1;
          if (__ballot(f1)) {
            if (f1) atomicMin(&s.s_fp[1], (unsigned long long)pos);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            cache1 = s.s_fp[1];
          }
          emit_lanes(s, c >= 2, 0xffffffffu, make_key(c, pos));
        }
      };
      one((uint32_t)lane < len, B.v[r]);
      if (len > 64) {
        cptr32 pp = (cptr32)(uintptr_t)s.part + (uint64_t)R.slot[r] * (s.np + 1) + p;     // (uniform: scalar load)
        const uint32_t *src = R.base[r] + pp[0];
        u32x4 nxt;
        __builtin_memcpy(&nxt, src + 64u + (uint32_t)lane * 4u, 16);       // (the array is padded: loads past the row's end are masked by index)
        for (uint32_t k0 = 64; k0 < len; k0 += 256) {
          const u32x4 cur = nxt;
          if (k0 + 256 < len) __builtin_memcpy(&nxt, src + k0 + 256u + (uint32_t)lane * 4u, 16);
          const uint32_t i = k0 + (uint32_t)lane * 4u;
#pragma unroll
          for (int j = 0; j < 4; ++j) one(i + j < len, cur[j]);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_s_waitcnt(0xc07f);            // every add has completed before the clears start
  }
}

#ifndef UGS_RANK_LEAN
#define UGS_RANK_LEAN 1
#endif
#ifndef UGS_RANK_NOAD
#define UGS_RANK_NOAD 0
#endif
template <int NR, bool LONG, bool LEANT = false>
__device__ __forceinline__ void process_batch(const ScanCtx &s, const Rows<NR> &R, const Batch<NR> &B, uint32_t p, unsigned long long &cache1, uint32_t &c1row)
{
  const int lane = s.lane;
  uint32_t *tbl = s.tbl;
  const uint32_t base_t = p * s.gsize;
  if (B.tail) {
    // a sub-row longer than a wavefront (rare at the chosen partition size): generic, row-ordered
    if constexpr (LONG) tail_mixed<NR>(s, R, B, p, base_t); else range_generic<4, false>(s, p, false, base_t, 0, 0, 0);
    return;
  }
  // Branch-free: lanes without a posting in row r aim at a private dummy word behind the table
  // (whatever they add / clear there is harmless), so the 2*NR LDS atomics issue back-to-back
  // and ONE wait covers the batch.
  // LDS byte address of a target's counter word = table base + (x >> 3) * 4 with x = target - base_t.  The base is
  // folded into x once per batch (x' = x + 2 * base, base a multiple of 4): address = (x' >> 1) & ~3, nibble = x' & 7.
  typedef uint32_t __attribute__((address_space(3))) *lds32;
  const uint32_t tb2 = 2u * (uint32_t)(uintptr_t)tbl;
  const uint32_t sub = base_t - tb2;                           // x' = posting - sub  (mod 2^32)
  const uint32_t dummy_x = ((s.tbl_words + (uint32_t)lane) << 3) + tb2;
  uint32_t ad[NR], sh[NR];   // LDS byte address and nibble shift of this lane's posting in row r
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const uint32_t x = (uint32_t)lane < B.len[r] ? B.v[r] - sub : dummy_x;
    ad[r] = (x >> 1) & ~3u;          // (UGS_RANK_NOAD: not kept - derived from the shift again in each phase, two fast VALU ops)
    sh[r] = x << 2;                    // only bits [4:0] are ever used (shift amounts, bit-field offset)
  }
#if UGS_RANK_NOAD
  auto adof = [&](int r) -> uint32_t { uint32_t t = sh[r]; asm volatile("" : "+v"(t)); return (t >> 3) & ~3u; };
#define ADR(r) adof(r)
#else
#define ADR(r) ad[r]
#endif
#pragma unroll
  for (int r = 0; r < NR; ++r) __hip_atomic_fetch_add((lds32)(uintptr_t)ADR(r), 1u << (sh[r] & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  // The clears below must find the final counts: every add of the batch has completed before the first clear issues
  // (explicit wait; it costs nothing measurable - the wave would wait for the clears' return values a few instructions
  // later anyway - and the result no longer rests on the order in which the LDS unit executes a wave's atomics).
  // tests/test_isa.py checks the emitted code for the wait.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  asm volatile("" ::: "memory");
  // ordered clears with return: LDS executes a wave's atomics in program order, so only the first
  // row that holds a target still sees its counter set and gets the target's final count back;
  // later rows of the same target read 0.
  uint32_t old[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) old[r] = __hip_atomic_fetch_and((lds32)(uintptr_t)ADR(r), ~(15u << (sh[r] & 31u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  uint32_t c[NR];            // count if this lane's posting is the first touch of its target, else 0
  uint32_t cnt = 0;          // rows in which this lane holds a first touch with count >= 2
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    // v_bfe_u32 takes the offset mod 32, so the unmasked shift (x << 2) serves directly
    c[r] = (uint32_t)lane < B.len[r] ? __builtin_amdgcn_ubfe(old[r], sh[r], 4u) : 0u;
    cnt += c[r] >= 2u ? 1u : 0u;
    if (s.small_path || (uint32_t)r <= c1row) {     // scalar test (c1row = row of the cached fp[1], kept in an SGPR): can this row still lower fp[1]?
      const uint32_t vr = (UGS_RANK_LEAN && LEANT) ? (sh[r] >> 2) + sub : B.v[r];          // LEAN: the posting is not kept, it follows from its shift
      const uint64_t pos = s.small_path ? (uint64_t)vr : (((uint64_t)r << 32) | vr);
      const bool f1 = c[r] == 1 && pos < cache1;
      if (__ballot(f1)) {
        if (f1) atomicMin(&s.s_fp[1], (unsigned long long)pos);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        cache1 = s.s_fp[1];
        c1row = __builtin_amdgcn_readfirstlane((uint32_t)(cache1 >> 32));
      }
    }
  }
  // every (lane,row) with count >= 2 of the whole partition is emitted with ONE slot allocation.  A lane almost
  // never holds more than one such row, so instead of walking the rows (NR predicated emit blocks per batch) each
  // lane first selects its lowest qualifying row with a chain of conditional moves and emits once; a second
  // item per lane takes another trip through the loop.
  if (__ballot(cnt != 0)) {
    const uint32_t incl = wave_incl_sum_u32(cnt);
    const uint32_t total = __builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t base = *s.wn;                                  // the wave's own segment: no atomic, no round trip
    *s.wn = base + total;
    uint32_t o = base + incl - cnt;
    auto emit = [&](uint32_t csel, uint32_t vsel, uint32_t rsel) {
      const uint64_t pos = s.small_path ? (uint64_t)vsel : (((uint64_t)rsel << 32) | vsel);
      put_key(s, o, make_key(csel, pos));
      ++o;
    };
    {
      uint32_t csel = 0, vsel = 0, rsel = 0;
#pragma unroll
      for (int r = NR - 1; r >= 0; --r) {
        const bool take = c[r] >= 2u;
        csel = take ? c[r] : csel; vsel = take ? ((UGS_RANK_LEAN && LEANT) ? sh[r] : B.v[r]) : vsel; rsel = take ? (uint32_t)r : rsel;
      }
      if (cnt) emit(csel, (UGS_RANK_LEAN && LEANT) ? (vsel >> 2) + sub : vsel, rsel);
      if (__ballot(cnt >= 2u)) {                       // rare: further items of a lane, rows above the one just emitted
#pragma unroll
        for (int r = 1; r < NR; ++r)
          if (c[r] >= 2u && (uint32_t)r > rsel) emit(c[r], (UGS_RANK_LEAN && LEANT) ? (sh[r] >> 2) + sub : B.v[r], (uint32_t)r);
      }
    }
  }
}

#undef ADR
template <int NR, bool LONG, bool SL = false>
__device__ __forceinline__ void scan_fast4(const ScanCtx &s)
{
  Rows<NR> R;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const uint32_t slot = __builtin_amdgcn_readfirstlane(s.s_slots[(uint32_t)r < s.ns ? r : 0]);
    R.slot[r] = slot;
    R.off[r] = slot * (s.np + 1u) * 4u;
    R.base[r] = s.postings + ((cptr64)(uintptr_t)s.row_off)[slot];
  }
  unsigned long long cache1 = KEY_INF;          // register copy of fp[1] (a stale-high filter)
  uint32_t c1row = 0xffffffffu;                 // its row, kept in an SGPR
  uint32_t p0 = s.wave;
  if (p0 >= s.np) return;
  const uint32_t last = s.np - 1;
  Batch<NR> A, B;
  issue_batch<NR, SL>(s, R, p0, A);
  for (;;) {
    const uint32_t p1 = p0 + s.wpb;
    issue_batch<NR, SL>(s, R, p1 < last ? p1 : last, B);
    process_batch<NR, LONG, SL>(s, R, A, p0, cache1, c1row);
    if (p1 >= s.np) break;
    const uint32_t p2 = p1 + s.wpb;
    issue_batch<NR, SL>(s, R, p2 < last ? p2 : last, A);
    process_batch<NR, LONG, SL>(s, R, B, p1, cache1, c1row);
    if (p2 >= s.np) break;
    p0 = p2;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// ---- mid-identity configuration: 16-255 sampled rows, 8-bit counters, table covers a whole partition.
// The register-resident scheme of the 4-bit path without its SGPR-resident row descriptors (too many rows): lane r
// holds row r's slot, row start and, per partition, its sub-row bounds (one vector load of the partition table for 64
// rows); a row's load address is made wave-uniform with three readlanes.  Rows are handled NR at a time: NR loads in
// flight, NR branch-free LDS atomics.  The postings are read twice (count, then ordered clear + extract); the second
// read hits the L2.  A partition with a sub-row longer than a wavefront goes through the generic code.
template <int NR, bool LONG>
__device__ __forceinline__ void scan_fast8(const ScanCtx &s)
{
  typedef uint32_t __attribute__((address_space(3))) *lds32;
  const int lane = s.lane;
  const uint32_t ns = s.ns;
  const uint32_t tb = (uint32_t)(uintptr_t)s.tbl;              // LDS byte address of the table (a multiple of 16)
  const uint32_t dummy_x = (s.tbl_words + (uint32_t)lane) * 4u;   // lanes without a posting aim at a private word behind the table
  unsigned long long cache1 = KEY_INF;
  for (uint32_t p = s.wave; p < s.np; p += s.wpb) {
    const uint32_t base_t = p * s.gsize;
    bool tail = false;
    uint32_t tot = 0, nrows = 0;
    uint64_t rstart0 = 0; uint32_t rlen0 = 0;                     // rows 0..63 stay in registers for both passes (the usual case: ns <= 64)
    for (uint32_t i0 = 0; i0 < ns; i0 += 64) {
      uint64_t rs = 0; uint32_t len = 0;
      if (i0 + lane < ns) {
        const uint32_t slot = s.s_slots[i0 + lane];
        const uint32_t *pp = s.part + (uint64_t)slot * (s.np + 1) + p;
        const uint32_t pa = pp[0];
        len = pp[1] - pa;
        if (i0 == 0) rs = s.row_off[slot] + pa;
      }
      if (i0 == 0) { rstart0 = rs; rlen0 = len; }
      tail = tail || __ballot(len > 64) != 0;
      tot += __builtin_amdgcn_readlane((int)wave_incl_sum_u32(len), 63);
      nrows += (uint32_t)__popcll(__ballot(len != 0));
    }
    // a sub-row longer than a wavefront, or sparse sub-rows (protein indexes: a handful of postings per row and
    // partition - the generic code flattens those row-major into full instructions): not this path's case
    if constexpr (LONG) if (tail) { range_long<8>(s, p, base_t); cache1 = s.s_fp[1]; continue; }
    if (tail || tot < 32u * nrows) { range_generic<8, false, true>(s, p, false, base_t, 0, 0, 0); continue; }
    for (int pass = 0; pass < 2; ++pass) {
      for (uint32_t i0 = 0; i0 < ns; i0 += 64) {
        uint64_t rstart = rstart0; uint32_t rlen = rlen0;          // lane r: first posting and length of row i0+r's sub-row
        if (i0) {
          rstart = 0; rlen = 0;
          if (i0 + lane < ns) {
            const uint32_t slot = s.s_slots[i0 + lane];
            const uint32_t *pp = s.part + (uint64_t)slot * (s.np + 1) + p;
            const uint32_t pa = pp[0];
            rstart = s.row_off[slot] + pa; rlen = pp[1] - pa;
          }
        }
        uint64_t rows = __ballot(rlen != 0);
        if (pass == 0) {
          // the count pass needs nothing but the postings: twice as many rows per trip, twice as many loads in flight
          constexpr int NC = 2 * NR;
          while (rows) {
            uint32_t v[NC], ln[NC];
#pragma unroll
            for (int u = 0; u < NC; ++u) {
              ln[u] = 0; v[u] = 0;
              if (rows) {
                const int r = __ffsll((long long)rows) - 1;
                rows &= rows - 1;
                const uint64_t st = shfl64(rstart, r);
                ln[u] = (uint32_t)__builtin_amdgcn_readlane((int)rlen, r);
                v[u] = (s.postings + st)[(uint32_t)lane];
              }
            }
#pragma unroll
            for (int u = 0; u < NC; ++u) {
              const uint32_t x = (uint32_t)lane < ln[u] ? v[u] - base_t : dummy_x;
              __hip_atomic_fetch_add((lds32)(uintptr_t)(tb + (x & ~3u)), 1u << ((x & 3u) << 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
          continue;
        }
        while (rows) {
          uint32_t v[NR], ad[NR], sh[NR], ln[NR], rix[NR];
#pragma unroll
          for (int u = 0; u < NR; ++u) {
            ln[u] = 0; rix[u] = 0; v[u] = 0;
            if (rows) {
              const int r = __ffsll((long long)rows) - 1;
              rows &= rows - 1;
              const uint64_t st = shfl64(rstart, r);
              ln[u] = (uint32_t)__builtin_amdgcn_readlane((int)rlen, r);
              rix[u] = i0 + (uint32_t)r;
              v[u] = (s.postings + st)[(uint32_t)lane];        // uniform base + lane: lanes beyond the sub-row read padding
            }
          }
#pragma unroll
          for (int u = 0; u < NR; ++u) {
            const uint32_t x = (uint32_t)lane < ln[u] ? v[u] - base_t : dummy_x;
            ad[u] = tb + (x & ~3u);
            sh[u] = (x & 3u) << 3;
          }
          // ordered clears with return (LDS executes a wave's atomics in program order): only the first row holding a
          // target sees its counter and gets the final count back, later rows read 0
          uint32_t old[NR];
#pragma unroll
          for (int u = 0; u < NR; ++u) old[u] = __hip_atomic_fetch_and((lds32)(uintptr_t)ad[u], ~(255u << sh[u]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          uint32_t c[NR], cnt = 0;
          const uint32_t c1hi = __builtin_amdgcn_readfirstlane((uint32_t)(cache1 >> 32));
#pragma unroll
          for (int u = 0; u < NR; ++u) {
            c[u] = (uint32_t)lane < ln[u] ? (old[u] >> sh[u]) & 255u : 0u;
            cnt += c[u] >= 2u ? 1u : 0u;
            if (s.small_path || rix[u] <= c1hi) {                // uniform: can this row still lower fp[1]?
              const uint64_t pos = s.small_path ? (uint64_t)v[u] : (((uint64_t)rix[u] << 32) | v[u]);
              const bool f1 = c[u] == 1 && pos < cache1;
              if (__ballot(f1)) {
                if (f1) atomicMin(&s.s_fp[1], (unsigned long long)pos);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                cache1 = s.s_fp[1];
              }
            }
          }
          if (__ballot(cnt != 0)) {
            const uint32_t incl = wave_incl_sum_u32(cnt);
            const uint32_t total = __builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t base = *s.wn;
            *s.wn = base + total;
            uint32_t o = base + incl - cnt;
#pragma unroll
            for (int u = 0; u < NR; ++u)
              if (c[u] >= 2u) {
                const uint64_t pos = s.small_path ? (uint64_t)v[u] : (((uint64_t)rix[u] << 32) | v[u]);
                put_key(s, o, make_key(c[u], pos));
                ++o;
              }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the count pass has completed before the ordered clears start
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

template <bool FILL, bool BATCH, bool FAST8, bool LONG, bool HOT = false>
__device__ __forceinline__ void scan_dispatch(const ScanCtx &s, int cb, uint32_t need, uint64_t fill_limit)
{
  if (cb == 4) {
    if (!FILL && s.ns <= 12 && s.tbl_words * 8 >= s.gsize) {
      if (s.ns <= 8) scan_fast4<8, LONG, HOT>(s); else if (s.ns <= 11) scan_fast4<11, LONG, HOT>(s); else scan_fast4<12, LONG, HOT>(s);
    }
    else scan_generic<4, FILL, BATCH, BATCH && !FAST8>(s, need, fill_limit);
  } else if (cb == 8) {
    if (!FILL && FAST8 && s.tbl_words * 4 >= s.gsize) scan_fast8<8, LONG>(s);
    else scan_generic<8, FILL, BATCH, BATCH && !FAST8>(s, need, fill_limit);
  }
  else scan_generic<16, FILL, BATCH, BATCH && !FAST8>(s, need, fill_limit);
}

// wave-uniform min of a u32 with DPP row shifts (no LDS crossbar): inclusive prefix-min inside each
// row of 16 lanes, then the four row results are combined on the scalar unit
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
  uint32_t x;
  x = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false); v = x < v ? x : v;   // row_shr:1
  x = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xf, 0xf, false); v = x < v ? x : v;   // row_shr:2
  x = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xf, 0xf, false); v = x < v ? x : v;   // row_shr:4
  x = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xf, 0xf, false); v = x < v ? x : v;   // row_shr:8
  const uint32_t a = __builtin_amdgcn_readlane((int)v, 15), b = __builtin_amdgcn_readlane((int)v, 31);
  const uint32_t c = __builtin_amdgcn_readlane((int)v, 47), d = __builtin_amdgcn_readlane((int)v, 63);
  const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
  return ab < cd ? ab : cd;
}
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v)
{
  const uint32_t hi = (uint32_t)(v >> 32);
  const uint32_t mh = wave_min_u32(hi);
  const uint32_t lo = hi == mh ? (uint32_t)v : 0xffffffffu;
  const uint32_t ml = wave_min_u32(lo);
  return ((uint64_t)mh << 32) | ml;
}

// Block-wide min of a 64-bit key (all threads get the result)
__device__ __forceinline__ uint64_t block_min_u64(uint64_t v, RankShared *sh, int wave, int wpb, int lane)
{
  for (int o = 32; o > 0; o >>= 1) {
    uint32_t lo = __shfl_xor((int)(uint32_t)v, o), hi = __shfl_xor((int)(uint32_t)(v >> 32), o);
    uint64_t x = ((uint64_t)hi << 32) | lo;
    v = x < v ? x : v;
  }
  __syncthreads();
  if (lane == 0) sh->red[wave] = v;
  __syncthreads();
  uint64_t r = sh->red[0];
  for (int w = 1; w < wpb; ++w) r = sh->red[w] < r ? sh->red[w] : r;
  return r;
}

__device__ __forceinline__ uint32_t block_sum_u32(uint32_t v, RankShared *sh, int wave, int wpb, int lane)
{
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor((int)v, o);
  __syncthreads();
  if (lane == 0) sh->wsum[wave] = v;
  __syncthreads();
  uint32_t r = 0;
  for (int w = 0; w < wpb; ++w) r += sh->wsum[w];
  return r;
}

#if UGS_RANK_TU != 1
// Chooses the sampled index rows of every unit (= query x strand) ahead of the scan: query letters -> UDB words
// (udbparams.cpp:540-555) -> unique words in first-occurrence order (udbsearcher.cpp:161-194) -> every step-th of them
// (GetWordCountingParams wordparams.cpp:179-191 via the host's step table).  One wavefront per unit, no block barriers:
// this stage is a chain of dependent loads with little arithmetic, so it runs at full occupancy in its own launch
// instead of stalling a 128-VGPR scan workgroup.
// cost class of a unit, heaviest first: 255 - ~9 classes per doubling of its postings
__device__ __forceinline__ uint32_t ugs_unit_cost_class(uint32_t cost)
{
  const uint32_t c = (uint32_t)(__log2f((float)cost + 1.0f) * 9.0f);
  return 255u - (c < 255u ? c : 255u);
}
// a unit's cost = the postings of its sampled rows (one thread per unit: at most a few dozen row lengths), and the histogram of the classes
__global__ __launch_bounds__(256) void k_unit_cost(UgsDbView db, UgsBatchView bv, uint32_t units, uint32_t ns_max)
{
  for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < units; u += gridDim.x * blockDim.x) {
    const uint32_t ns = bv.unit_ns[u];
    unsigned long long sum = 0;
    for (uint32_t r = 0; r < ns; ++r) { const uint32_t slot = bv.unit_slots[(uint64_t)u * ns_max + r]; sum += db.row_off[slot + 1] - db.row_off[slot]; }
    const uint32_t cost = sum > 0xffffffffull ? 0xffffffffu : (uint32_t)sum;
    bv.unit_cost[u] = cost;
    atomicAdd(&bv.order_hist[ugs_unit_cost_class(cost)], 1u);
  }
}
// units by descending cost class (a counting sort: every workgroup sums the histogram for itself, positions inside a class by atomics -
// the order inside a class does not matter)
__global__ __launch_bounds__(256) void k_unit_order(UgsBatchView bv, uint32_t units)
{
  __shared__ uint32_t s_base[256];
  {
    uint32_t sum = 0;
    for (uint32_t k = 0; k < threadIdx.x; ++k) sum += bv.order_hist[k];
    s_base[threadIdx.x] = sum;
  }
  __syncthreads();
  for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < units; u += gridDim.x * blockDim.x) {
    const uint32_t b = ugs_unit_cost_class(bv.unit_cost[u]);
    const uint32_t pos = s_base[b] + atomicAdd(&bv.order_hist[256 + b], 1u);
    bv.unit_order[pos] = u;
  }
}

__global__ __launch_bounds__(256) void k_rank_setup(UgsDbView db, UgsBatchView bv, uint32_t ns_max)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wpb = blockDim.x >> 6;      // (k_rank_setup: the wave index in an SGPR measured SLOWER here, 1.82 -> 2.01 ms on C2, r5)
  const uint32_t maxq = (bv.max_qlen + 15u) & ~15u;
  uint8_t *s_udb = smem, *s_sc = smem + 256;                       // letter -> index letter / alignment score code (k_align's s_sc over its class table)
  unsigned char *wb = smem + 512 + (size_t)wave * ((size_t)maxq * 8 + 2048);
  uint32_t *s_words = (uint32_t *)wb;
  uint8_t *s_q = wb + (size_t)maxq * 4, *s_first = s_q + maxq;
  uint16_t *s_dup = (uint16_t *)(s_first + maxq);                 // positions whose word may have occurred before
  uint32_t *s_hc = (uint32_t *)(wb + (size_t)maxq * 8);            // hash counts
  const UgsTables *tab = db.tab;
  for (int k = tid; k < 256; k += blockDim.x) {
    s_udb[k] = tab->udb_letter[k];
    const uint32_t cl = tab->cls[k] & 31u;
    s_sc[k] = (cl < 26u && tab->udb_letter['A' + cl] != 0xff) ? tab->hsp_letter['A' + cl] : 4;
  }
  __syncthreads();
  const uint32_t units = bv.nq * bv.nstrand;
  const int W = db.word_len;
  const bool small_path = !db.big;
  unsigned long long psum = 0;
  (void)wpb;
  // units are handed out dynamically in chunks of 16 per wave (one same-address atomic per unit would take longer than
  // this whole kernel)
  for (uint32_t unit = 0, chunk_end = 0;; ++unit) {
    if (unit >= chunk_end) {
      uint32_t c0 = 0;
      if (lane == 0) c0 = (uint32_t)atomicAdd(&bv.counters[UGS_CTR_NEXT_SETUP], 16ull);
      unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0);
      chunk_end = unit + 16;
    }
    if (unit >= units) break;
    const uint32_t qi = unit / bv.nstrand, strand = unit % bv.nstrand;
    const uint64_t qo = bv.qoffs[qi];
    const uint32_t L = (uint32_t)(bv.qoffs[qi + 1] - qo);
    // ---- query letters (reverse-complemented for strand 1: seqinfo.cpp:292-323)
    for (uint32_t p = lane; p < L; p += 64) s_q[p] = strand == 0 ? bv.qseqs[qo + p] : tab->comp[bv.qseqs[qo + (L - 1 - p)]];
    __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory");
    if (bv.qpk) {
      // ---- the letters packed for the alignment stage: word k = letters 16 k .. 16 k + 15, 2 bits each (score code & 3), and in the same
      // layout one bit per letter that is not A/C/G/T/U (pack_codes of ugs_align.hip); zero behind the last letter, three zero words more
      uint2 *qp = bv.qpk + (uint64_t)unit * bv.qpk_stride;
      const uint32_t nw = (L + 15) >> 4;
      for (uint32_t k = lane; k < bv.qpk_stride; k += 64) {
        uint32_t v = 0, iv = 0;
        if (k < nw) {
          const uint4 d4 = *(const uint4 *)(s_q + 16 * k);
          const uint32_t d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const uint32_t pos = 16u * k + 4u * (uint32_t)q + (uint32_t)e;
              const uint32_t sc = pos < L ? (uint32_t)s_sc[(d[q] >> (8 * e)) & 0xffu] : 0u;
              v |= (sc & 3u) << (2 * (4 * q + e)); iv |= (sc >> 2) << (2 * (4 * q + e));
            }
        }
        qp[k] = make_uint2(v, iv);
      }
    }
    for (uint32_t p = lane; p < ((L + 3) & ~3u); p += 64) {
      uint32_t w = UGS_BAD_WORD;
      if (p + W <= L) {
        uint32_t acc = 0; bool ok = true;
        for (int k = 0; k < W; ++k) { uint32_t l = s_udb[s_q[p + k]]; ok = ok && (l != 0xff); acc = acc * db.alpha + l; }
        if (ok) w = acc;
      }
      s_words[p] = w;
    }
    __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory");
    // ---- first occurrences: any earlier position with the same word?  A 1024-bucket hash count tells most positions apart at
    // once (a word alone in its bucket occurs once); only the others are compared with the earlier positions (128-bit LDS
    // reads), and those are listed densely first so that the scans fill whole wavefronts
    for (uint32_t k = lane; k < 512; k += 64) s_hc[k] = 0;                      // 1024 16-bit counters
    __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory");
    for (uint32_t p = lane; p < L; p += 64) {
      const uint32_t w = s_words[p];
      if (w != UGS_BAD_WORD) { const uint32_t h = (w ^ (w >> 10)) & 1023u; atomicAdd(&s_hc[h >> 1], 1u << ((h & 1u) * 16)); }
    }
    __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory");
    uint32_t ndup = 0;
    for (uint32_t p0 = 0; p0 < L; p0 += 64) {
      const uint32_t p = p0 + lane;
      bool need = false;
      if (p < L) {
        const uint32_t w = s_words[p];
        const uint32_t h = (w ^ (w >> 10)) & 1023u;
        const bool valid = w != UGS_BAD_WORD;
        need = valid && ((s_hc[h >> 1] >> ((h & 1u) * 16)) & 0xffffu) > 1u;
        s_first[p] = valid ? 1 : 0;                                               // (corrected below for the listed ones)
      }
      const uint64_t m = __ballot(need);
      if (need) s_dup[ndup + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)p;
      ndup += (uint32_t)__popcll(m);
    }
    __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory");
    if (ndup <= 512u) {
      // every earlier occurrence of a listed position's word is listed too (same bucket): the listed words, densely (in the hash
      // counters' array, which is done), are all a listed position is compared with - a fifth of the scan over all earlier positions
      uint32_t *s_dw = s_hc;
      for (uint32_t i = lane; i < ndup; i += 64) s_dw[i] = s_words[s_dup[i]];
      __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory");
      for (uint32_t i = lane; i < ndup; i += 64) {
        const uint32_t w = s_dw[i];
        const uint4 *v4 = (const uint4 *)s_dw;
        const uint32_t nq4 = i >> 2;
        bool dupf = false;
        for (uint32_t q4 = 0; q4 < nq4; ++q4) { const uint4 x = v4[q4]; dupf = dupf | (x.x == w) | (x.y == w) | (x.z == w) | (x.w == w); }
        for (uint32_t q = nq4 << 2; q < i; ++q) dupf = dupf | (s_dw[q] == w);
        if (dupf) s_first[s_dup[i]] = 0;
      }
    } else
    for (uint32_t i = lane; i < ndup; i += 64) {
      const uint32_t p = s_dup[i], w = s_words[p];
      const uint4 *v4 = (const uint4 *)s_words;
      const uint32_t nq4 = p >> 2;
      bool dupf = false;
      for (uint32_t q4 = 0; q4 < nq4; ++q4) { const uint4 x = v4[q4]; dupf = dupf | (x.x == w) | (x.y == w) | (x.z == w) | (x.w == w); }
      for (uint32_t q = nq4 << 2; q < p; ++q) dupf = dupf | (s_words[q] == w);
      if (dupf) s_first[p] = 0;
    }
    __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory");
    uint32_t Nu = 0;
    for (uint32_t p0 = 0; p0 < L; p0 += 64) Nu += (uint32_t)__popcll(__ballot(p0 + lane < L && s_first[p0 + lane]));
    uint32_t step = 1;
    if (!small_path) step = db.step_tab[Nu < db.step_n ? Nu : db.step_n - 1];
    const uint32_t ns_q = Nu == 0 ? 0 : (Nu + step - 1) / step;
    const uint32_t ns = ns_q <= ns_max ? ns_q : ns_max;
    if (ns_q > ns_max && lane == 0) atomicOr(&bv.counters[UGS_CTR_ERR], (unsigned long long)UGS_ERR_NS);
    __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory");
    // ---- ranks of unique words -> sampled slots (every step-th unique word)
    uint32_t run = 0;
    uint32_t *out = bv.unit_slots + (uint64_t)unit * ns_max;
    for (uint32_t p0 = 0; p0 < L; p0 += 64) {
      const uint32_t p = p0 + lane;
      const bool f = p < L && s_first[p];
      const uint64_t m = __ballot(f);
      if (f) {
        const uint32_t rank = run + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (rank % step == 0 && rank / step < ns) {
          const uint32_t slot = s_words[p];
          out[rank / step] = slot;
          psum += db.row_off[slot + 1] - db.row_off[slot];          // algorithmic postings P(q)
        }
      }
      run += (uint32_t)__popcll(m);
    }
    if (lane == 0) bv.unit_ns[unit] = ns;
    __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory");
  }
  for (int o = 32; o > 0; o >>= 1) psum += __shfl_xor((long long)psum, o);
  if (lane == 0 && psum) atomicAdd(&bv.counters[UGS_CTR_POSTINGS], psum);
}

#endif   // UGS_RANK_TU != 1

// Big path, fewer than K targets with count >= 2 and MinValue <= 1: the count-1 targets in first-touch order are the postings of
// row 0 in ascending target order, then those of row 1 that no earlier row holds, ...  A target has count 1 exactly when it is not
// one of the count >= 2 targets, and those are all selected already (cand[0 .. nsel), one per lane).  So the fill is a walk over
// the first few postings of the first row(s) against that list - no second scan.  One wave; returns the new number of candidates.
// (Inlined like every helper of this kernel: out-of-line functions cost it dearly, see scan_generic's history in DESIGN.md.  The dense
// 8-bit kernels do not use it, see the call.)
__device__ __forceinline__ uint32_t big_path_fill(const uint64_t *row_off, const uint32_t *postings, const uint32_t *s_slots, uint32_t ns,
                                                            uint32_t *cand, uint32_t *cand_cnt, uint64_t *cand_key, uint32_t K, uint32_t nsel, int lane)
{
  const uint32_t mine = (uint32_t)lane < nsel ? cand[lane] : 0xffffffffu;
  uint32_t filled = nsel;
  for (uint32_t r = 0; r < ns && filled < K; ++r) {
    const uint32_t slot = s_slots[r];
    const uint64_t ra = row_off[slot], rb = row_off[slot + 1];
    for (uint64_t k0 = ra; k0 < rb && filled < K; k0 += 64) {
      const bool on = k0 + (uint64_t)lane < rb;
      const uint32_t t = on ? postings[k0 + lane] : 0u;
      bool in_set = false;
      for (uint32_t j = 0; j < nsel; ++j) in_set = in_set || t == (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)j);
      const bool e = on && !in_set;
      const uint64_t m = __ballot(e);
      const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if (e && filled + rank < K) {
        cand[filled + rank] = t;
        cand_cnt[filled + rank] = 1u;
        if (cand_key) cand_key[filled + rank] = make_key(1, ((uint64_t)r << 32) | t);
      }
      const uint32_t n = (uint32_t)__popcll(m);
      filled = filled + n < K ? filled + n : K;
    }
  }
  return filled;
}

// SMALL: the database is at or below -big (small ranking path) - a per-database constant, so the two rankers are two
// instantiations and each carries only its own branches through the hot loop
// BATCH: launches whose tables are wider than 4 bits (16-255 sampled rows: mid-identity searches) walk the generic path
// for every unit; their row loops keep four loads in flight.  The 4-bit launches keep the code the hot path was tuned with.
// FAST8: dense indexes (nucleotide) take scan_fast8 for 8-bit tables; sparse dictionaries (protein) keep the generic code,
// which flattens their short sub-rows - again an instantiation of its own, so that neither pays for the other's registers
// LONG: databases with very long index rows (an abundant family shares its words) - the partitions whose sub-rows exceed a
// wavefront then go through range_long; again an instantiation of its own (see range_long)
template <bool SMALL, bool BATCH, bool FAST8, bool LONG, bool WIDE = false>
#ifndef UGS_RANK_WGS
#define UGS_RANK_WGS 4
#endif
#ifndef UGS_RANK_WGS_HOT
#define UGS_RANK_WGS_HOT 6
#endif
#ifndef UGS_RANK_WGS_LONG
#define UGS_RANK_WGS_LONG 4
#endif
#ifndef UGS_ELDS_LONG
#define UGS_ELDS_LONG UGS_ELDS
#endif
#ifndef UGS_ELDS_BATCH
#define UGS_ELDS_BATCH 128u       // 8- and 16-bit counter kernels: thousands of keys per unit, nearly all of them in the HBM part anyway
#endif
__global__ __launch_bounds__(256, (!SMALL && !BATCH && !FAST8 && !WIDE) ? (LONG ? UGS_RANK_WGS_LONG : UGS_RANK_WGS_HOT) : UGS_RANK_WGS) void k_rank(UgsDbView db, UgsBatchView bv, uint32_t ns_max, uint32_t tbl_words, uint32_t part_words)
{
  constexpr bool HOT = !SMALL && !BATCH && !FAST8 && !LONG;       // Big path, 4-bit counters, uniform rows: five workgroups per CU (issue_batch<.., true>)
  constexpr bool HOTL = !SMALL && !BATCH && !FAST8 && LONG;     // the same path on a skewed database (cluster_fast's centroids)
  constexpr uint32_t ELDS = HOT ? UGS_ELDS_HOT : (HOTL ? UGS_ELDS_LONG : (BATCH ? UGS_ELDS_BATCH : UGS_ELDS));
#ifndef UGS_RANK_SL_LONG
#define UGS_RANK_SL_LONG 1
#endif
  // the asm partition-table loads (issue_batch<.., true>) address the table and the rows with 32-bit byte offsets; WIDE: the instantiations
  // for an index whose partition table reaches 4 GiB or whose longest row reaches 2^30 postings keep the compiler's 64-bit address
  // arithmetic (issue_batch<.., false>) - slower, never selected for the BASELINE shapes (ugs_host.cpp plan_launch)
  constexpr bool SLOAD = !WIDE && (HOT || (UGS_RANK_SL_LONG && !SMALL && !BATCH && !FAST8 && LONG));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wpb = nthr >> 6;      // wave index in an SGPR
  const uint32_t maxq = (bv.max_qlen + 15u) & ~15u;
  // LDS carve (all offsets multiples of 16)
  size_t off = 0;
  RankShared *sh = (RankShared *)(smem + off); off += (sizeof(RankShared) + 15) & ~(size_t)15;
  unsigned long long *s_fp = (unsigned long long *)(smem + off); off += (((size_t)ns_max + 1) * 8 + 15) & ~(size_t)15;
  uint64_t *s_ebuf = (uint64_t *)(smem + off); off += (size_t)ELDS * 8;
  uint32_t *s_slots = (uint32_t *)(smem + off); off += (((size_t)ns_max) * 4 + 15) & ~(size_t)15;
  uint32_t *s_ev_c = (uint32_t *)(smem + off); off += (((size_t)ns_max + 1) * 4 + 15) & ~(size_t)15;     // -bump events
  uint32_t *s_ev_minu = (uint32_t *)(smem + off); off += (((size_t)ns_max + 1) * 4 + 15) & ~(size_t)15;
  // per-wave selections (+8 pad).  HOT: they live in wave 0's counter table, which is idle (and clean) between a unit's scan and the
  // next unit's - the words used are zeroed again when the unit is done; 2 KB less LDS is what lets a sixth workgroup fit a CU
  constexpr uint32_t WSEL_WORDS = (4 * UGS_KMAX + 8) * 2;
  uint64_t *s_wsel = (uint64_t *)(smem + off); if (!HOT) off += (size_t)WSEL_WORDS * 4;
  uint32_t *s_part = (uint32_t *)(smem + off); off += (size_t)part_words * 4;       // cached partition-table rows of the sampled words
  uint32_t *tbl = (uint32_t *)(smem + off) + (size_t)wave * (tbl_words + 64);      // +64 dummy words per wave
  if (HOT) s_wsel = (uint64_t *)(smem + off);                                       // (= the first waves' tables, WSEL_WORDS * 4 = 2112 bytes: plan_launch checks that the tables are that large; off is a multiple of 16)

  const UgsTables *tab = db.tab;
  // Big path: behind a bitmap kernel (ugs_rank2.hip) this kernel takes the units that one deferred, from its list
  const bool deferred = !SMALL && bv.use_defer != 0;
  const uint32_t units = deferred ? (uint32_t)bv.counters[bv.use_defer == 2u ? UGS_CTR_DEFER2 : UGS_CTR_DEFER] : bv.nq * bv.nstrand;
  const uint32_t K = bv.K;
  const int W = db.word_len;
  constexpr bool small_path = SMALL;
  uint64_t *ebuf = bv.emit_buf + (uint64_t)blockIdx.x * bv.emit_cap;
  const uint64_t ecap = bv.emit_cap;

  // counter tables must start clean; pass 2 restores that invariant after every partition
  for (uint32_t k = lane; k < tbl_words + 64; k += 64) tbl[k] = 0;

  unsigned long long tacc0 = 0, tacc1 = 0, tacc2 = 0, tacc3 = 0;
  // units are handed out dynamically (one atomic per unit, fetched a unit ahead by thread 0)
  uint32_t next_unit = 0;
  if (tid == 0) sh->pad1 = (uint32_t)atomicAdd(&bv.counters[UGS_CTR_NEXT_RANK], 1ull);
  __syncthreads();
  for (uint32_t uq = sh->pad1; uq < units; uq = sh->pad1) {
    // (the list is taken from its END: the bitmap kernel appends a unit when it gives it up, the units it worked on longest - the heaviest,
    // the ones that should not be started last - come last)
    // (... unless the units were handed out heaviest first, cluster_fast: then the list starts with the heavy ones)
    const uint32_t unit = deferred ? (bv.unit_order ? bv.defer_list[uq] : bv.defer_list[units - 1u - uq]) : (bv.unit_order ? bv.unit_order[uq] : uq);
    __syncthreads();                     // everyone has read sh->pad1
    const unsigned long long tk0 = clock64();
    // ---- the sampled index rows of this unit were chosen by k_rank_setup
    if (tid == 0) { sh->emit_n = 0; sh->n_sel = 0; sh->last_key = 0; sh->ncl = 0; }
    for (uint32_t k = tid; k < 256; k += nthr) sh->hist[k] = 0;
    for (uint32_t c = tid; c <= ns_max; c += nthr) s_fp[c] = KEY_INF;
    const uint32_t ns = __builtin_amdgcn_readfirstlane(bv.unit_ns[unit]);
    // (a query without a valid word has ns == 0: slot 0 stands in so that the row descriptors below never read an
    // uninitialised LDS word; every sub-row then has length 0 and nothing is ranked)
    for (uint32_t i = tid; i < (ns ? ns : 1u); i += nthr) s_slots[i] = ns ? bv.unit_slots[(uint64_t)unit * ns_max + i] : 0u;
    __syncthreads();
    // ---- the scan; counter width by the largest possible count (= ns)
    const int cb0 = ns <= 15 ? 4 : (ns <= 255 ? 8 : 16);
    ScanCtx sc;
    sc.pq = nullptr;
    sc.row_off = db.row_off; sc.part = db.part; sc.postings = db.postings; sc.s_slots = s_slots; sc.tbl = tbl;
    sc.s_part = nullptr;
    sc.hist = (cb0 == 4 && !small_path) ? sh->hist : nullptr;
    sc.s_fp = s_fp; sc.sh = sh; sc.ecap = ecap; sc.ns = ns; sc.np = db.np; sc.gsize = db.gsize;
    sc.elw = ELDS / (uint32_t)wpb; sc.ecapw = ecap / (uint64_t)wpb;
    sc.s_ebuf = s_ebuf + (size_t)wave * sc.elw; sc.ebuf = ebuf + (uint64_t)wave * sc.ecapw;
    uint32_t wave_n = 0;
    sc.wn = &wave_n;
    sc.tbl_words = tbl_words; sc.wave = wave; sc.wpb = wpb; sc.lane = lane; sc.small_path = small_path;
    const int cb = cb0;
    const unsigned long long tk1 = clock64();
    scan_dispatch<false, BATCH, FAST8, LONG, SLOAD>(sc, cb, 0, 0);
    // the next unit's index is fetched here: late enough to stay out of the scan's register budget, early enough
    // for the atomic's latency to hide behind the selection
    if (tid == 0) next_unit = (uint32_t)atomicAdd(&bv.counters[UGS_CTR_NEXT_RANK], 1ull);
    const unsigned long long tk2w = clock64();
    if (lane == 0) sh->wn[wave] = wave_n;
    __threadfence_block();
    __syncthreads();
    const unsigned long long tk2 = clock64();
    // ---- the emitted entries: wave w's segment holds wn[w] keys; entry k of the unit = segment w, place k - woff[w]
    const uint32_t elw = sc.elw; const uint64_t ecapw = sc.ecapw;
    uint32_t woff1 = 0, woff2 = 0, woff3 = 0, n_emit = 0;
    auto read_counts = [&]() {
      // (a wave that emitted more than its share holds only the first elw + ecapw keys: the unit is flagged below and searched again
      // with a larger buffer; until then nothing past the stored keys is read)
      const uint32_t wcap = (uint32_t)std::min<uint64_t>((uint64_t)elw + ecapw, 0xffffffffull);
      const uint32_t n0 = min(sh->wn[0], wcap), n1 = wpb > 1 ? min(sh->wn[1], wcap) : 0u, n2 = wpb > 2 ? min(sh->wn[2], wcap) : 0u, n3 = wpb > 3 ? min(sh->wn[3], wcap) : 0u;
      woff1 = n0; woff2 = n0 + n1; woff3 = n0 + n1 + n2; n_emit = n0 + n1 + n2 + n3;
    };
    read_counts();
    auto load_key = [&](uint32_t k) -> uint64_t {
      const uint32_t w = (uint32_t)(k >= woff1) + (uint32_t)(k >= woff2) + (uint32_t)(k >= woff3);
      const uint32_t i = k - (w == 0 ? 0u : w == 1 ? woff1 : w == 2 ? woff2 : woff3);
      return i < elw ? s_ebuf[(size_t)w * elw + i] : ebuf[(uint64_t)w * ecapw + (i - elw)];
    };
    {   // first position per count value and, on the 4-bit Big path, the (count, first row) class sizes - from the keys
      // (count >= 2 entries only; fp[1] is maintained by the scan)
      auto pass_a = [&](uint64_t key) {
        const uint32_t cc = key_count(key);
        // (nearly every key has count 2: a plain read first - after the first few keys hardly any position is still below the
        // minimum, and the atomics on that one word no longer queue up)
        const unsigned long long pos = (unsigned long long)(key & POS_MASK);
        if (pos < *(volatile unsigned long long *)&s_fp[cc]) atomicMin(&s_fp[cc], pos);
        if (sc.hist) atomicAdd(&sh->hist[cc * 16 + ((uint32_t)(key >> 32) & 0xfu)], 1u);
      };
      if constexpr (BATCH || LONG) {
        // (LONG: skewed databases emit as many)
        // mid-identity launches emit ~10 k keys per unit, most of them in the HBM part of the buffer: four loads in flight per
        // thread instead of one (eight: the scan_fast8 loop of the same kernel spills, 162 -> 185 ms) (a loop over single loads waits out the memory latency once per key)
        for (uint32_t k0 = tid; k0 < n_emit; k0 += 4 * nthr) {
          uint64_t kk[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { const uint32_t k = k0 + (uint32_t)u * nthr; kk[u] = k < n_emit ? load_key(k) : KEY_INF; }
#pragma unroll
          for (int u = 0; u < 4; ++u) if (kk[u] != KEY_INF) pass_a(kk[u]);
        }
      } else
        for (uint32_t k = tid; k < n_emit; k += nthr) pass_a(load_key(k));
      __syncthreads();
    }

    // ---- cut-offs.  NextValue = running max just before the max last increased, in scan
    // (first-touch / ascending-target) order  == max{c < M : fp[c] < fp[M]}  (countsort.cpp:13-24,114-126)
    if (wave == 0) {
      uint32_t m = 0;
      for (uint32_t c = lane + 1; c <= ns; c += 64) if (s_fp[c] != KEY_INF) m = c > m ? c : m;
      for (int o = 32; o > 0; o >>= 1) { uint32_t x = __shfl_xor((int)m, o); m = x > m ? x : m; }
      uint32_t nv = 0;
      if (m) {
        const unsigned long long fpm = s_fp[m];
        for (uint32_t c = lane + 1; c < m; c += 64) if (s_fp[c] < fpm) nv = c > nv ? c : nv;
        for (int o = 32; o > 0; o >>= 1) { uint32_t x = __shfl_xor((int)nv, o); nv = x > nv ? x : nv; }
      }
      if (lane == 0) {
        sh->M = m; sh->next_value = nv; sh->min_value = nv / 2;
        sh->nev = 0; sh->fill_limit = KEY_INF;
        if (bv.cl_ev) {
          // cluster_fast (ugs_cluster.cpp): the greedy loop's host side merges this unit's walk with the centroids founded
          // inside the same batch, which can move the CountSort cut-off (countsort.cpp:13-24) either way.  So the candidates
          // are chosen WITHOUT the MinValue cut-off here, and the strict prefix maxima of the scan (count, first position) -
          // all the cut-off depends on - go to the host, which applies the cut-off of the merged scan.
          sh->min_value = 0;
          unsigned long long sufmin = KEY_INF;
          uint32_t ne = 0;
          for (uint32_t c = m; c >= 1; --c) {
            const unsigned long long f = s_fp[c];
            if (f != KEY_INF && f < sufmin) { if (ne < UGS_CL_EV) bv.cl_ev[(uint64_t)unit * UGS_CL_EV + ne] = ((uint64_t)c << POS_BITS) | f; ++ne; sufmin = f; }
          }
          bv.cl_info[(uint64_t)unit * 4 + 0] = m; bv.cl_info[(uint64_t)unit * 4 + 1] = nv; bv.cl_info[(uint64_t)unit * 4 + 2] = ne;
        }
        if (small_path && m && db.bump_pct != 0) {
          // strict prefix maxima in ascending-target order = counts c whose first position
          // precedes the first position of every larger count (udbusortedsearcher.cpp:230-267)
          unsigned long long sufmin = KEY_INF;
          uint32_t nev = 0;
          for (uint32_t c = m; c >= 1; --c) {
            const unsigned long long f = s_fp[c];
            if (f != KEY_INF && f < sufmin) { s_ev_c[nev++] = c; sufmin = f; }   // descending c
          }
          // replay -bump over the events in scan order (ascending c); record MinU after each
          const double Bump = db.bump_pct / 100.0;
          uint32_t MinU = 1, MaxCount = 0;
          for (int e = (int)nev - 1; e >= 0; --e) {
            const uint32_t n = s_ev_c[e];
            const uint32_t NewMin = (uint32_t)(n * Bump);
            if (NewMin > MinU && NewMin < MaxCount) MinU = NewMin;
            MaxCount = n;
            s_ev_minu[e] = MinU;
            if (MinU > 1 && sh->fill_limit == KEY_INF) sh->fill_limit = s_fp[n];
          }
          sh->nev = nev;
        }
      }
    }
    __syncthreads();
    const uint32_t min_value = sh->min_value;
    const uint32_t nev = sh->nev;

    // kept(entry): count >= MinValue and (small path) count >= MinU in force at its position
    PairQ pairq;
    pairq.mask = SMALL ? (db.pair_mask & ~(uint32_t)UGS_P_SELFID) : 0u;
    if (SMALL && pairq.mask) {
      const uint32_t qi = unit / bv.nstrand;
      pairq.ql = (uint32_t)(bv.qoffs[qi + 1] - bv.qoffs[qi]);
      pairq.qkey = bv.q_key ? bv.q_key[qi] : 0u; pairq.qsize = bv.q_size ? bv.q_size[qi] : 0xffffffffu;
      pairq.min_sizeratio = db.min_sizeratio; pairq.minqt = db.minqt; pairq.maxqt = db.maxqt; pairq.minsl = db.minsl; pairq.maxsl = db.maxsl;
      pairq.offs = db.offs; pairq.t_key = db.t_key; pairq.t_size = db.t_size;
      sc.pq = &pairq;                                        // the count-1 fill pass drops refused targets as it emits
    }
    auto kept = [&](uint64_t key) -> bool {
      const uint32_t c = key_count(key);
      if (c < min_value) return false;
      if (nev) {
        const uint64_t pos = key & POS_MASK;
        uint32_t minu = 1;
        // events are stored by descending count == descending position; MinU in force at pos
        // = MinU after the last event strictly before pos
        for (uint32_t e = 0; e < nev; ++e) if (s_fp[s_ev_c[e]] < pos) { minu = s_ev_minu[e]; break; }
        if (c < minu) return false;
      }
      if (SMALL && pairq.mask && pair_reject(pairq, key_target(key))) return false;
      return true;
    };

    for (int phase = 0; phase < 2; ++phase) {
      const uint32_t N = n_emit;
      // select the K smallest kept keys by repeated min above the previous one
      uint32_t nsel = sh->n_sel;
      uint64_t last = sh->last_key;
      bool exhausted = false;
      bool done_fast = false;
      if (phase == 0 && sc.hist && sh->M >= 2) {
        // ---- class-histogram selection (4-bit Big path).  Classes in key order q = (M-c)*16 + i;
        // a prefix sum over the class sizes gives the class that holds the K-th key, so only the
        // few entries up to that class are ranked (all-pairs inside one wave) - no extraction rounds.
        const uint32_t Mx = sh->M;
        const uint32_t cmin = min_value > 2 ? min_value : 2;
        if (wave == 0) {
          uint32_t loc[4], sum = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t q = lane * 4 + e, cc = Mx - (q >> 4);
            loc[e] = (q >> 4) < Mx && cc >= cmin ? sh->hist[cc * 16 + (q & 15)] : 0u;
            sum += loc[e];
          }
          uint32_t incl = sum;
          for (int o = 1; o < 64; o <<= 1) { uint32_t x = __shfl_up((int)incl, o); if (lane >= o) incl += x; }
          uint32_t run = incl - sum, qc = 0xffffffffu;
#pragma unroll
          for (int e = 0; e < 4; ++e) { run += loc[e]; if (qc == 0xffffffffu && run >= K) qc = lane * 4 + e; }
          const uint64_t mk = __ballot(qc != 0xffffffffu);
          uint32_t qcut = 0xfffffffeu;                         // fewer than K entries: take them all
          if (mk) qcut = __builtin_amdgcn_readlane((int)qc, __ffsll((long long)mk) - 1);
          if (lane == 0) sh->qcut = qcut;
        }
        __syncthreads();
        const uint32_t qcut = sh->qcut;
        // gather every kept entry whose class is not after the cut class
        auto gather_one = [&](uint64_t key, bool valid) {
          bool take = false;
          if (valid) {
            const uint32_t cc = key_count(key), ii = (uint32_t)(key >> 32) & 0xfffu;
            take = cc >= cmin && ((Mx - cc) * 16 + ii) <= qcut;
          }
          const uint64_t mk = __ballot(take);
          if (mk) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&sh->ncl, (uint32_t)__popcll(mk));
            base = __builtin_amdgcn_readfirstlane(base);
            const uint32_t slot = base + __popcll(mk & ((1ull << lane) - 1ull));
            if (take && slot < 4 * UGS_KMAX) s_wsel[slot] = key;
          }
        };
        if constexpr (LONG) {
          for (uint32_t k0 = 0; k0 < N; k0 += 4 * nthr) {       // four key loads in flight per thread
            uint64_t kk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const uint32_t k = k0 + (uint32_t)u * nthr + tid; kk[u] = k < N ? load_key(k) : KEY_INF; }
#pragma unroll
            for (int u = 0; u < 4; ++u) gather_one(kk[u], kk[u] != KEY_INF);
          }
        } else
          for (uint32_t k0 = 0; k0 < N; k0 += nthr) {
            const uint32_t k = k0 + tid;
            gather_one(k < N ? load_key(k) : 0, k < N);
          }
        __syncthreads();
        const uint32_t ncl = sh->ncl;
        if (ncl <= 64u * (uint32_t)wpb && ncl <= 4 * UGS_KMAX) {
          if ((uint32_t)tid < 8u && ncl + (uint32_t)tid < 4 * UGS_KMAX + 8) s_wsel[ncl + tid] = KEY_INF;      // padding for the 8-wide ranking loop
          __syncthreads();
          // all-pairs ranking: wave w ranks entries [64w, 64w+64) against the whole list (LDS broadcast reads)
          const uint32_t me = wave * 64 + lane;
          const uint64_t mykey = me < ncl ? s_wsel[me] : KEY_INF;
          uint32_t rank = 0;
          if (wave * 64u < ncl) {
            // 128-bit LDS broadcast reads, 8 keys in flight per trip: the list is padded to a multiple of 8 with
            // KEY_INF (never below a real key), so the loop has no tail and the loads of one trip are independent
            const uint4 *w4 = (const uint4 *)s_wsel;
            const uint32_t n8 = (ncl + 7u) & ~7u;
            for (uint32_t j = 0; j < n8; j += 8) {
              const uint4 x0 = w4[(j >> 1) + 0], x1 = w4[(j >> 1) + 1], x2 = w4[(j >> 1) + 2], x3 = w4[(j >> 1) + 3];
              rank += ((((uint64_t)x0.y << 32) | x0.x) < mykey) + ((((uint64_t)x0.w << 32) | x0.z) < mykey);
              rank += ((((uint64_t)x1.y << 32) | x1.x) < mykey) + ((((uint64_t)x1.w << 32) | x1.z) < mykey);
              rank += ((((uint64_t)x2.y << 32) | x2.x) < mykey) + ((((uint64_t)x2.w << 32) | x2.z) < mykey);
              rank += ((((uint64_t)x3.y << 32) | x3.x) < mykey) + ((((uint64_t)x3.w << 32) | x3.z) < mykey);
            }
          }
          const uint32_t nout = ncl < K ? ncl : K;
          if (me < ncl && rank < K) {
            bv.cand[(uint64_t)unit * K + rank] = key_target(mykey);
            bv.cand_cnt[(uint64_t)unit * K + rank] = key_count(mykey);
            if (bv.cand_key) bv.cand_key[(uint64_t)unit * K + rank] = mykey;
            if (rank + 1 == nout) sh->last_key = mykey;
          }
          if (tid == 0) { sh->n_sel = nout; sh->exhausted = ncl < K ? 1u : 0u; }
          __syncthreads();
          nsel = sh->n_sel; last = sh->last_key; exhausted = sh->exhausted != 0;
          done_fast = true;
        }
      }
      if (done_fast) {
      } else if (N <= 64u * SELQ * (uint32_t)wpb) {
        // every wave selects the K smallest of its own interleaved share (entries in registers,
        // DPP reductions, no barriers), then wave 0 merges the wpb*K survivors
        {
          uint64_t ent[SELQ];
#pragma unroll
          for (int e = 0; e < SELQ; ++e) {
            const uint32_t k = (lane + e * 64) * wpb + wave;
            uint64_t key = KEY_INF;
            if (k < N) { key = load_key(k); if (!kept(key) || !((nsel == 0 && last == 0) || key > last)) key = KEY_INF; }
            ent[e] = key;
          }
          uint64_t llast = 0; bool lfirst = true;
          const uint32_t want = K - nsel;
          for (uint32_t j = 0; j < want; ++j) {
            uint64_t best = KEY_INF;
#pragma unroll
            for (int e = 0; e < SELQ; ++e) { const uint64_t key = ent[e]; if ((lfirst || key > llast) && key < best) best = key; }
            best = wave_min_u64(best);
            if (lane == 0) s_wsel[wave * UGS_KMAX + j] = best;
            if (best == KEY_INF) { for (uint32_t j2 = j + 1 + lane; j2 < want; j2 += 64) s_wsel[wave * UGS_KMAX + j2] = KEY_INF; break; }
            llast = best; lfirst = false;
          }
        }
        __syncthreads();
        if (wave == 0) {
          const uint32_t want = K - nsel;
          uint64_t m[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t idx = lane + e * 64, w2 = idx / UGS_KMAX, j2 = idx % UGS_KMAX;
            m[e] = (w2 < (uint32_t)wpb && j2 < want) ? s_wsel[w2 * UGS_KMAX + j2] : KEY_INF;
          }
          bool mfirst = true; uint64_t mlast = 0;
          while (nsel < K) {
            uint64_t best = KEY_INF;
#pragma unroll
            for (int e = 0; e < 4; ++e) if ((mfirst || m[e] > mlast) && m[e] < best) best = m[e];
            best = wave_min_u64(best);
            if (best == KEY_INF) { exhausted = true; break; }
            if (lane == 0) {
              bv.cand[(uint64_t)unit * K + nsel] = key_target(best);
              bv.cand_cnt[(uint64_t)unit * K + nsel] = key_count(best);
              if (bv.cand_key) bv.cand_key[(uint64_t)unit * K + nsel] = best;
            }
            mlast = best; mfirst = false; last = best; ++nsel;
          }
          if (lane == 0) { sh->n_sel = nsel; sh->last_key = last; sh->exhausted = exhausted ? 1u : 0u; }
        }
        __syncthreads();
        nsel = sh->n_sel; last = sh->last_key; exhausted = sh->exhausted != 0;
      } else if (BATCH || LONG) {
        // (LONG: a skewed database - cluster_fast's centroids - emits ~10 k keys per unit as well; without this branch the
        // fallback below makes K passes over them)
        // ---- many emitted entries (mid-identity searches: ~1 % of the database shares two of 40 sampled words): radix
        // select.  Byte by byte from the top of the key, a 256-bin histogram of the entries still in the running for
        // the K-th place narrows the class that holds it; when the entries below the class plus the class itself fit
        // the ranking buffer they are gathered and ranked all-pairs.  5 passes over the entries instead of K.
        const uint32_t want = K - nsel;
        const uint32_t cap = 4 * UGS_KMAX < 64u * (uint32_t)wpb ? 4 * UGS_KMAX : 64u * (uint32_t)wpb;
        auto eligible = [&](uint64_t key) -> bool { return ((nsel == 0 && last == 0) || key > last) && kept(key); };
        // (r03: this barrier was missing.  A LONG 4-bit kernel comes here from the class-histogram attempt above, whose last shared read
        // is `ncl = sh->ncl` - behind ITS last barrier.  Without a barrier here thread 0 could reset sh->ncl before a slower wave had read
        // it; that wave then saw 0, took the "fits the ranking buffer" branch with its own barriers, and the workgroup's waves ran
        // different code: cluster_fast C3 gave a different clustering in 1 of 40 runs with the round-2 kernels, in 1 of 6 with the
        // faster round-3 ones (tools/cluster_repeat.py).)
        __syncthreads();
        if (tid == 0) { sh->red[0] = 0; sh->red[1] = 0; sh->qcut = 0; sh->pad2 = 0xffffffffu; sh->pad3 = 0; sh->ncl = 0; }
        __syncthreads();
        for (int shift = 48; shift >= 0; shift -= 8) {
          for (uint32_t k = tid; k < 256; k += nthr) sh->hist[k] = 0;
          __syncthreads();
          const uint64_t pref = sh->red[0], pmask = sh->red[1];
          for (uint32_t k0 = tid; k0 < N; k0 += 4 * nthr) {       // four keys in flight per thread
            uint64_t kk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const uint32_t k = k0 + (uint32_t)u * nthr; kk[u] = k < N ? load_key(k) : KEY_INF; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (kk[u] != KEY_INF && (kk[u] & pmask) == pref && eligible(kk[u])) atomicAdd(&sh->hist[(uint32_t)(kk[u] >> shift) & 255u], 1u);
          }
          __syncthreads();
          if (wave == 0) {
            const uint32_t nb = sh->qcut, rem = want - nb;      // entries surely selected so far / still to come from this class
            uint32_t loc[4], sum = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) { loc[e] = sh->hist[lane * 4 + e]; sum += loc[e]; }
            uint32_t incl = sum;
            for (int o = 1; o < 64; o <<= 1) { uint32_t x = __shfl_up((int)incl, o); if (lane >= o) incl += x; }
            uint32_t run = incl - sum, dsel = 0xffffffffu, before = 0, csz = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) { if (dsel == 0xffffffffu && run + loc[e] >= rem) { dsel = lane * 4 + e; before = run; csz = loc[e]; } run += loc[e]; }
            const uint64_t mk = __ballot(dsel != 0xffffffffu);
            if (mk) {
              const int Ls = __ffsll((long long)mk) - 1;
              if (lane == Ls) {
                sh->qcut = nb + before; sh->pad2 = csz;
                sh->red[0] = pref | ((uint64_t)dsel << shift); sh->red[1] = pmask | (255ull << shift);
              }
            } else if (lane == 0) { sh->pad3 = 1; sh->pad2 = __builtin_amdgcn_readlane((int)incl, 63); }   // fewer than `want` entries in all: take them all
          }
          __syncthreads();
          if (sh->pad3 || sh->qcut + sh->pad2 <= cap) break;
        }
        {
          const uint64_t pref = sh->red[0], pmask = sh->red[1];
          for (uint32_t k0 = 0; k0 < N; k0 += 4 * nthr) {
            uint64_t kk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const uint32_t k = k0 + (uint32_t)u * nthr + tid; kk[u] = k < N ? load_key(k) : KEY_INF; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const uint64_t key = kk[u];
              const bool take = key != KEY_INF && (key & pmask) <= pref && eligible(key);
              const uint64_t mk = __ballot(take);
              if (mk) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&sh->ncl, (uint32_t)__popcll(mk));
                base = __builtin_amdgcn_readfirstlane(base);
                const uint32_t slot = base + __popcll(mk & ((1ull << lane) - 1ull));
                if (take && slot < 4 * UGS_KMAX) s_wsel[slot] = key;
              }
            }
          }
          __syncthreads();
          const uint32_t ncl = sh->ncl < cap ? sh->ncl : cap;
          if ((uint32_t)tid < 8u && ncl + (uint32_t)tid < 4 * UGS_KMAX + 8) s_wsel[ncl + tid] = KEY_INF;
          __syncthreads();
          const uint32_t me = wave * 64 + lane;
          const uint64_t mykey = me < ncl ? s_wsel[me] : KEY_INF;
          uint32_t rank = 0;
          if (wave * 64u < ncl) {
            const uint4 *w4 = (const uint4 *)s_wsel;
            const uint32_t n8 = (ncl + 7u) & ~7u;
            for (uint32_t j = 0; j < n8; j += 8) {
              const uint4 x0 = w4[(j >> 1) + 0], x1 = w4[(j >> 1) + 1], x2 = w4[(j >> 1) + 2], x3 = w4[(j >> 1) + 3];
              rank += ((((uint64_t)x0.y << 32) | x0.x) < mykey) + ((((uint64_t)x0.w << 32) | x0.z) < mykey);
              rank += ((((uint64_t)x1.y << 32) | x1.x) < mykey) + ((((uint64_t)x1.w << 32) | x1.z) < mykey);
              rank += ((((uint64_t)x2.y << 32) | x2.x) < mykey) + ((((uint64_t)x2.w << 32) | x2.z) < mykey);
              rank += ((((uint64_t)x3.y << 32) | x3.x) < mykey) + ((((uint64_t)x3.w << 32) | x3.z) < mykey);
            }
          }
          const uint32_t nout = ncl < want ? ncl : want;
          if (me < ncl && rank < want) {
            bv.cand[(uint64_t)unit * K + nsel + rank] = key_target(mykey);
            bv.cand_cnt[(uint64_t)unit * K + nsel + rank] = key_count(mykey);
            if (bv.cand_key) bv.cand_key[(uint64_t)unit * K + nsel + rank] = mykey;
            if (rank + 1 == nout) sh->last_key = mykey;
          }
          __syncthreads();
          if (tid == 0) { sh->n_sel = nsel + nout; sh->exhausted = ncl < want ? 1u : 0u; }
          __syncthreads();
          nsel = sh->n_sel; last = sh->last_key; exhausted = sh->exhausted != 0;
        }
      } else {
        while (nsel < K) {
          uint64_t best = KEY_INF;
          for (uint32_t k = tid; k < N; k += nthr) {
            const uint64_t key = load_key(k);
            if (((nsel == 0 && last == 0) || key > last) && key < best && kept(key)) best = key;
          }
          best = block_min_u64(best, sh, wave, wpb, lane);
          if (best == KEY_INF) { exhausted = true; break; }
          if (tid == 0) {
            bv.cand[(uint64_t)unit * K + nsel] = key_target(best);
            bv.cand_cnt[(uint64_t)unit * K + nsel] = key_count(best);
            if (bv.cand_key) bv.cand_key[(uint64_t)unit * K + nsel] = best;
          }
          last = best; ++nsel;
        }
        __syncthreads();
        if (tid == 0) { sh->n_sel = nsel; sh->last_key = last; }
        __syncthreads();
      }
      if (phase == 1 || !exhausted || nsel >= K) break;
      // fewer than K candidates with count >= 2: append count-1 targets in scan order if the
      // cut-offs keep them (MinValue <= 1; small path additionally position < first MinU bump)
      if (min_value > 1 || sh->M == 0) break;
      if constexpr (!SMALL && !FAST8) {       // (the FAST8 kernels - mid-identity nt - all but never come here and keep the scan: with this code compiled in their scan_fast8 loop spills, 171 -> 190 ms)
        // Big path: the count-1 targets in first-touch order are the postings of row 0 in ascending target order, then those of
        // row 1 that no earlier row holds, ...  A target has count 1 exactly when it is not one of the count >= 2 targets, and
        // those are all selected already (the list is exhausted and MinValue <= 1 keeps every one of them): fewer than K, one per
        // lane.  So the fill is a walk over the first few postings of the first row(s) against that list - no second scan.
        if (wave == 0) {
          const uint32_t filled = big_path_fill(db.row_off, db.postings, s_slots, ns, bv.cand + (uint64_t)unit * K, bv.cand_cnt + (uint64_t)unit * K,
                                                bv.cand_key ? bv.cand_key + (uint64_t)unit * K : nullptr, K, nsel, lane);
          if (lane == 0) sh->n_sel = filled;
        }
        __syncthreads();
        break;
      }
      const uint32_t need = K - nsel;
      const uint64_t fill_limit = sh->fill_limit;
      __syncthreads();
      scan_dispatch<true, BATCH, FAST8, LONG>(sc, cb, need, fill_limit);
      if (lane == 0) sh->wn[wave] = wave_n;
      __threadfence_block();
      __syncthreads();
      read_counts();
    }
    tacc0 += tk1 - tk0; tacc1 += tk2w - tk1; tacc2 += tk2 - tk2w; tacc3 += clock64() - tk2;
    if (tid == 0) {
      bool over = false;
      for (int w2 = 0; w2 < wpb; ++w2)
        if ((uint64_t)sh->wn[w2] > (uint64_t)elw + ecapw) {      // the host grows the buffer to the demand and runs the search again
          atomicOr(&bv.counters[UGS_CTR_ERR], (unsigned long long)UGS_ERR_EMIT);
          atomicMax(&bv.counters[UGS_CTR_EMIT_MAX], (unsigned long long)sh->wn[w2]);
          over = true;
        }
      bv.cand_n[unit] = over ? 0u : sh->n_sel;                   // (no walk over a list chosen from truncated keys)
      sh->pad1 = next_unit;
    }
    if (HOT) for (uint32_t k = tid; k < WSEL_WORDS; k += nthr) ((uint32_t *)s_wsel)[k] = 0;     // wave 0's table is a counter table again
    __syncthreads();
  }
  if (tid == 0) {
    atomicAdd(&bv.counters[UGS_CTR_T0], tacc0); atomicAdd(&bv.counters[UGS_CTR_T1], tacc1);
    atomicAdd(&bv.counters[UGS_CTR_T2], tacc2); atomicAdd(&bv.counters[UGS_CTR_T3], tacc3);
  }
}

// ---- two translation units from this one file (usearch12_amd/build.py):
//   UGS_RANK_TU == 1: nothing but the HOT instantiation k_rank<false,false,false,false> (C2 / C4), compiled with LLVM's
//                     "iterative-maxocc" machine scheduler: 54.5 ms against 55.7 ms with the default scheduler on C2, the C4 shard 148
//                     against 155 ms - while the mid-identity kernels lose 10 % under the same option (135 against 122 ms)
//   UGS_RANK_TU == 2: every other instantiation, the setup kernel and the host side; the HOT kernel is an extern template here
//   undefined / 0:    everything in one unit (tools/build_variant.sh)
#ifndef UGS_RANK_TU
#define UGS_RANK_TU 0
#endif
#if UGS_RANK_TU == 1
template __global__ void k_rank<false, false, false, false>(UgsDbView, UgsBatchView, uint32_t, uint32_t, uint32_t);
#else
#if UGS_RANK_TU == 2
extern template __global__ void k_rank<false, false, false, false>(UgsDbView, UgsBatchView, uint32_t, uint32_t, uint32_t);
#endif
// resident workgroups per CU for a given block size / dynamic LDS (VGPR- and LDS-limited): the
// persistent grid must not exceed it, or the surplus workgroups run as a second, unbalanced round
// ordinal of the instantiation rank_kernel() picks (0 .. UGS_RANK_INSTANCES - 1): the library records which ones a process launched
// (ugs_rank_instances_seen; the test-suite asserts at its end that every compiled instantiation ran)
static const void *rank_kernel(int big, int bits, int fast8, int longrows, int wide = 0, int *ordinal = nullptr)
{
  const int mode = bits == 4 ? 0 : (fast8 ? 2 : 1);
  int dummy; int &id = ordinal ? *ordinal : dummy;
#ifdef UGS_ONLY_HOT               // tuning builds (tools/build_variant.sh): only the C2 instantiation, compiles in seconds
  (void)big; (void)mode; (void)longrows; (void)wide;
  id = UGS_RI_BIG4;
  return (const void *)k_rank<false, false, false, false>;
#else
  if (wide && big && mode == 0) {  // (only the two Big-path 4-bit kernels have 32-bit offsets to outgrow)
    id = longrows ? UGS_RI_BIG4_LONG_WIDE : UGS_RI_BIG4_WIDE;
    return longrows ? (const void *)k_rank<false, false, false, true, true> : (const void *)k_rank<false, false, false, false, true>;
  }
  if (longrows) {
    id = big ? (mode == 0 ? UGS_RI_BIG4_LONG : mode == 1 ? UGS_RI_BIG_FLAT : UGS_RI_BIG_DENSE_LONG) : (mode == 0 ? UGS_RI_SMALL4_LONG : mode == 1 ? UGS_RI_SMALL_FLAT : UGS_RI_SMALL_DENSE_LONG);
    return big ? (mode == 0 ? (const void *)k_rank<false, false, false, true> : mode == 1 ? (const void *)k_rank<false, true, false, false> : (const void *)k_rank<false, true, true, true>)
               : (mode == 0 ? (const void *)k_rank<true, false, false, true> : mode == 1 ? (const void *)k_rank<true, true, false, false> : (const void *)k_rank<true, true, true, true>);
  }
  id = big ? (mode == 0 ? UGS_RI_BIG4 : mode == 1 ? UGS_RI_BIG_FLAT : UGS_RI_BIG_DENSE) : (mode == 0 ? UGS_RI_SMALL4 : mode == 1 ? UGS_RI_SMALL_FLAT : UGS_RI_SMALL_DENSE);
  return big ? (mode == 0 ? (const void *)k_rank<false, false, false, false> : mode == 1 ? (const void *)k_rank<false, true, false, false> : (const void *)k_rank<false, true, true, false>)
             : (mode == 0 ? (const void *)k_rank<true, false, false, false> : mode == 1 ? (const void *)k_rank<true, true, false, false> : (const void *)k_rank<true, true, true, false>);
#endif
}
// bit i: instantiation i (UGS_RANK_INST_TABLE, ugs_dev.h) was launched by this process
static std::atomic<unsigned long long> g_rank_seen{0};
unsigned long long ugs_rank_instances_seen(unsigned long long *compiled)
{
#define UGS_RI_BIT(i, n, s) | (1ull << (i))
#ifdef UGS_ONLY_HOT
  if (compiled) *compiled = (1ull << UGS_RI_BIG4) | (1ull << UGS_RI_R2) | (1ull << UGS_RI_R2G) | (1ull << UGS_RI_R2_CL) | (1ull << UGS_RI_R3G) | (1ull << UGS_RI_R2_P16) | (1ull << UGS_RI_R2_HV);
#else
  if (compiled) *compiled = 0ull UGS_RANK_INST_TABLE(UGS_RI_BIT);
#endif
#undef UGS_RI_BIT
  return g_rank_seen.load();
}
const char *ugs_rank_instance_name(int ordinal)
{
#define UGS_RI_NAME(i, n, s) if (ordinal == (i)) return s;
  UGS_RANK_INST_TABLE(UGS_RI_NAME)
#undef UGS_RI_NAME
  return nullptr;
}
// the instantiation with five workgroups per CU and a smaller LDS key segment (must mirror k_rank's HOT)
int ugs_rank_is_hot(int big, int bits, int fast8, int longrows) { (void)fast8; return bits != 4 ? 3 : (big ? (longrows ? 2 : 1) : 0); }   // 1 = HOT, 2 = its LONG twin, 3 = wider counters (BATCH kernels)

int ugs_rank_blocks_per_cu(int threads, size_t lds, int big, int bits, int fast8, int longrows, int wide)
{
  int n = 0;
  const void *fn = rank_kernel(big, bits, fast8, longrows, wide);      // (the instantiations differ in their register budget: the one that will run is asked)
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, threads, lds) != hipSuccess || n < 1) n = 1;
  return n;
}

// LDS bytes of everything in k_rank's carve except the per-wave counter tables (must mirror the kernel)
size_t ugs_rank_fixed_lds(uint32_t ns_max, uint32_t max_qlen, uint32_t part_words, int hot)
{
  const size_t maxq = (max_qlen + 15u) & ~15u;
  size_t off = 0;
  off += (sizeof(RankShared) + 15) & ~(size_t)15;
  off += (((size_t)ns_max + 1) * 8 + 15) & ~(size_t)15;        // s_fp
  off += (size_t)(hot == 1 ? UGS_ELDS_HOT : (hot == 2 ? UGS_ELDS_LONG : (hot == 3 ? UGS_ELDS_BATCH : UGS_ELDS))) * 8;            // s_ebuf
  off += (((size_t)ns_max) * 4 + 15) & ~(size_t)15;            // s_slots
  off += 2 * ((((size_t)ns_max + 1) * 4 + 15) & ~(size_t)15);  // s_ev_c, s_ev_minu
  if (hot != 1) off += ((size_t)4 * UGS_KMAX + 8) * 8;         // s_wsel (+8 pad; HOT: inside wave 0's counter table)
  off += (size_t)part_words * 4;                               // s_part
  return (off + 15) & ~(size_t)15;
}

int ugs_launch_rank(const UgsDbView &db, const UgsBatchView &b, const UgsRankLaunch &L, hipStream_t st_rank, hipEvent_t ev_setup_done,
                    const UgsRank2Params *r2, int r2_grid, hipEvent_t ev_r2_done, hipStream_t st_setup, hipEvent_t ev_rank_start)
{
  const bool split = st_setup != nullptr && st_setup != st_rank && ev_setup_done != nullptr;
  hipStream_t st = split ? st_setup : st_rank;                  // (stage 1 below; the ranking kernels run on st_rank)
  const uint32_t tbl_words = (uint32_t)(((uint64_t)db.gsize * L.bits) / 32);
  dim3 grid(L.grid), block(64 * L.wpb);
  {   // stage 1: sampled rows of every unit (one wavefront per unit, as many workgroups as fit)
    const uint32_t units = b.nq * b.nstrand, maxq = (b.max_qlen + 15u) & ~15u;
    const size_t slds = 512 + 4 * ((size_t)maxq * 8 + 2048);
    if (slds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void *)k_rank_setup, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds));
    int per_cu = 0, ncu = 0, dev = 0;
    HIPCHK(hipGetDevice(&dev));
    HIPCHK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_rank_setup, 256, slds) != hipSuccess || per_cu < 1) per_cu = 1;
    const uint32_t sgrid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((units + 3) / 4, (uint64_t)ncu * per_cu));
    if (units && b.unit_cost) HIPCHK(hipMemsetAsync(b.order_hist, 0, 512 * sizeof(uint32_t), st));
    if (units) hipLaunchKernelGGL(k_rank_setup, dim3(sgrid), dim3(256), slds, st, db, b, L.ns_max);
    HIPCHK(hipGetLastError());
    if (units && b.unit_cost) {
      hipLaunchKernelGGL(k_unit_cost, dim3(std::min<uint32_t>((units + 255) / 256, 256u)), dim3(256), 0, st, db, b, units, L.ns_max);
      hipLaunchKernelGGL(k_unit_order, dim3(std::min<uint32_t>((units + 255) / 256, 64u)), dim3(256), 0, st, b, units);
      HIPCHK(hipGetLastError());
    }
    if (ev_setup_done) HIPCHK(hipEventRecord(ev_setup_done, st));
    if (split) HIPCHK(hipStreamWaitEvent(st_rank, ev_setup_done, 0));
    st = st_rank;
    if (ev_rank_start) HIPCHK(hipEventRecord(ev_rank_start, st));
    if (L.debug_sync) { HIPCHK(hipStreamSynchronize(st)); fprintf(stderr, "[ugs] k_rank_setup done (grid %u, lds %zu); k_rank grid %d x %d lds %zu bits %d ns_max %u tbl_words %u\n", sgrid, slds, L.grid, L.wpb, L.lds, L.bits, L.ns_max, tbl_words); }
  }
#ifdef UGS_ONLY_HOT
  if (!(db.big && L.bits == 4 && !L.longrows)) { ugs_set_error("UGS_ONLY_HOT build"); return UGS_E_ENVELOPE; }
#endif
  // dense Big-path index: the bitmap kernel ranks the units and lists the ones outside its envelope, which k_rank (below) then takes
  if (r2) {
    if (r2->gather == 2u) RCCHK_(ugs_launch_rank3g(db, b, *r2, r2_grid, st));
    else RCCHK_(ugs_launch_rank2(db, b, *r2, r2_grid, st));
    if (ev_r2_done) HIPCHK(hipEventRecord(ev_r2_done, st));
  }
  int ordinal = 0;
  const void *fn = rank_kernel(db.big, L.bits, L.fast8, L.longrows, L.wide, &ordinal);
  g_rank_seen.fetch_or((1ull << ordinal) | (r2 ? (1ull << (r2->gather == 2u ? UGS_RI_R3G : r2->gather ? UGS_RI_R2G : (b.cand_key ? UGS_RI_R2_CL : (r2->post16 ? UGS_RI_R2_P16 : UGS_RI_R2)))) : 0ull) |
                       ((r2 && r2->hv_grid > 0 && r2->defer2 && b.cand_key && !r2->gather) ? (1ull << UGS_RI_R2_HV) : 0ull));
  HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds));
  {
    UgsDbView a0 = db; UgsBatchView a1 = b; uint32_t a2 = L.ns_max, a3 = tbl_words, a4 = L.part_words;
    a1.use_defer = r2 ? 1u : 0u;
    if (r2 && r2->hv_grid > 0 && r2->defer2 && b.cand_key && !r2->gather) { a1.use_defer = 2u; a1.defer_list = r2->defer2; }      // (behind the heavy-unit kernel)
    void *args[] = {&a0, &a1, &a2, &a3, &a4};
    if (ugs_kernel_log) ugs_before_launch("k_rank");
    HIPCHK(hipLaunchKernel(fn, grid, block, args, L.lds, st));
    if (ugs_kernel_log) ugs_after_launch("k_rank", st);
  }
  HIPCHK(hipGetLastError());
  return UGS_OK;
}
#endif   // UGS_RANK_TU != 1
