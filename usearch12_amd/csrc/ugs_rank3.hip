// ugs_rank3.hip - k_rank3g: the Big-path ranking kernel for SPARSE indexes (protein dictionaries), second design (r5), gfx950.
//
// Replaces the same reference code as k_rank2g / k_rank's Big path:
//   UDBUsortedSearcher::UDBSearchBig  scan + first-touch list            udbusortedsearcherbig.cpp:82-100
//   CountSortSubsetDesc (+ the NextValue / MinValue = prevMax/2 cut-off)   countsort.cpp:110-191
//
// What a protein query looks like (C5: 300 aa vs 2 M sequences): ~ 59 sampled rows of ~ 200 postings = 12 k postings spread over
// 2 M targets, of which a few dozen targets are touched twice (chance) and one - the relative - a dozen times.  k_rank2g finds them
// EXACTLY with one bit per target, which forces 31 partitions of 65 536 targets and, per partition, a descriptor layout, three
// quarter-full chunks, a grouping and a bitmap reset: its time is per-partition overhead (DESIGN.md section 3 "K-rank2g": 0.18 of HBM peak).
//
// k_rank3g asks the cheaper question first.  The target space is cut into a few SUPER-partitions (whole partitions of the index's
// partition table, as many per unit as keep ~ 4 096 postings in each: C5 three of ~ 700 k targets); per super-partition the rows' segments
// are streamed TWICE, in any order:
//   pass 1  every posting goes through a blocked two-bit filter in LDS (4 KB: word = target bits 5..14, one bit from the target's low
//           five bits, one from a multiplicative hash; ONE ds_or_rtn per posting).  A posting that finds both its bits set is a
//           SUSPECT: a target with count >= 2 is always one (its second posting finds the bits its first one set), a false suspect
//           costs time only (a few dozen per super-partition).
//   pass 2  the filter is zeroed, the suspects' exact bits (target mod 32 768) are set, and the same segments are streamed again
//           (from the L2 / MALL now): every posting whose bit is set leaves a RECORD (row, target) - ALL occurrences of every
//           target with count >= 2, plus the single occurrences of false suspects and of their aliases.
//   then    the few dozen records are grouped by target through a 512-entry hash table in the same LDS (compare-and-swap insertion,
//           the rows of a target's records as a 64-bit mask): count = its bits, first-touch row = the lowest - no scan order is needed -,
//           keys, pruning and the kept-key list exactly as k_rank2g, and the same end of the unit (r2g_finish_unit).
// A chunk of the stream = 8 groups of 8 lanes, a group = 8 consecutive 16-byte quads of ONE row's segment (a "group-step"; the list
// of a super-partition's group-steps is laid out once - a prefix sum over the row lanes - and serves both passes).  The loads run
// through the accumulator-register ring of k_rank2 (four chunks in flight, waits counted by hand).
// A super-partition that overflows the group-step list (R3_GSCAP) or the records (R3_RCAP) is scanned again as two halves.  Outside the
// envelope (more than 63 sampled rows, a SINGLE partition that overflows - an abundant family -, a full kept-key list) the unit is
// DEFERRED to k_rank like k_rank2g's.  Never a different result, never a CPU path.
#include "ugs_dev.h"
#include "ugs_rank2.h"
#include <cstdlib>
#include <cstdio>
#include <algorithm>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ugs_set_error("%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return UGS_E_HIP; } } while (0)

#include "ugs_ring_dev.h"

#define R3_BW 1024u             // 32-bit words of the filter (pass 1) = the suspects' bitmap (pass 2) = the grouping table
#define R3_GSCAP 256u           // group-steps (8 quads = up to 32 postings of one row) of one super-partition; more: it is halved
#define R3_GSPAD 40u            // empty group-steps behind the list (the ring's look-ahead reads them instead of testing the index)
#define R3_RCAP 256u            // records (pass 1: suspects, pass 2: occurrences) of one super-partition; more: it is halved
#define R3_TAB 512u             // entries of the grouping table: 32-bit keys (2 KB) | 64-bit row masks (4 KB) over the filter AND the group-step list
#define R3_HMUL 0x9E3779u       // 24-bit multiplier of the filter's second (and third) bit
#ifndef R3_FBITS
#define R3_FBITS 2              // bits a posting sets in its filter word (3: fewer false suspects, two more instructions per posting - C5 shape 4.51 vs 4.39 ms per 200 k queries)
#endif
#if R3_FBITS == 3
#define R3_THIRD_BIT(h) | (1u << (((h) >> 5) & 31u))
#else
#define R3_THIRD_BIT(h)
#endif
#ifdef R3_DEFER_STATS          // profiling counters: T0 / T1 suspects / records, T2 group-steps, T3 super-partitions scanned, T4 most suspects of one; why
                               // units are deferred: T5 (suspects), T6 (records), T7 (group-step list or kept keys)
#define R3_WHY(c) do { if (lane == 0) atomicAdd(&bv.counters[c], 1ull); } while (0)
#define R3_STAT(c, v) do { if (lane == 0) atomicAdd(&bv.counters[c], (unsigned long long)(v)); } while (0)
#define R3_STATMAX(c, v) do { if (lane == 0) atomicMax(&bv.counters[c], (unsigned long long)(v)); } while (0)
#else
#define R3_WHY(c) do { } while (0)
#define R3_STAT(c, v) do { } while (0)
#define R3_STATMAX(c, v) do { } while (0)
#endif
#ifdef R3_CLOCKS2               // + the posting waits of both passes (in T5)
#define R3_CLK2(...) __VA_ARGS__
#else
#define R3_CLK2(...)
#endif
#ifdef R3_CLOCKS
#define R3_CLK(...) __VA_ARGS__
#else
#define R3_CLK(...)
#endif

__host__ __device__ constexpr uint32_t r3_fixed_bytes() { return R3_BW * 4u + (R3_GSCAP + R3_GSPAD) * 8u + (R3_RCAP + 64u) * 4u + 64u * 3u * 4u; }

__global__ __launch_bounds__(64, 4) void k_rank3g(UgsDbView db, UgsBatchView bv, UgsRank2Params prm)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x;
  const uint32_t np = prm.np, K = bv.K, kcap = prm.kcap;
  // ---- LDS carve (r3_fixed_bytes + the kept keys)
  uint32_t *s_bm = (uint32_t *)smem;                                    // [R3_BW] filter | suspects' bitmap | grouping table
  uint2 *s_gs = (uint2 *)(smem + R3_BW * 4u);                           // [R3_GSCAP + R3_GSPAD] group-steps of the super-partition
  uint32_t *s_rec = (uint32_t *)(s_gs + R3_GSCAP + R3_GSPAD);           // [R3_RCAP + 64] records (+ slack)
  uint32_t *s_sel = s_rec;                                              //   after the scan: [64] selected targets (for the fill)
  uint64_t *s_fpk = (uint64_t *)(s_rec + 64);                           //   after the scan: [64] smallest key per count value
  uint32_t *s_c2 = s_rec + R3_RCAP + 64;                                // [64] kept count-2 keys per row
  uint32_t *s_cum = s_c2 + 64;                                          // [64] ... with that row or a lower one
  uint32_t *s_slots = s_cum + 64;                                       // [64] sampled slots of the unit (by row)
  uint64_t *s_kl = (uint64_t *)(s_slots + 64);                          // [kcap + 2] kept keys
  const uint32_t units = bv.nq * bv.nstrand;
  const uint32_t ns_max = prm.ns_max;
  const uint32_t *postings = db.postings;
  const uint32_t grp = lane >> 3, lo16 = (lane & 7u) * 16u, i0 = (lane & 7u) * 4u;
  unsigned long long n_done_local = 0;
  R3_CLK(unsigned long long tc[5] = {0, 0, 0, 0, 0};)
  R3_CLK2(unsigned long long tw = 0;)

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
    const uint32_t ns = (uint32_t)__builtin_amdgcn_readfirstlane((int)bv.unit_ns[unit]);
    bool bad = ns > R2G_MAXROWS;
    if (ns == 0) { if (lane == 0) bv.cand_n[unit] = 0; continue; }
    uint32_t nk = 0;
    bool any_posting = false;
    if (!bad) {
      // ---- row lanes: lane r < ns owns sampled row r (no scan order is needed: counts and lowest rows come from the records)
      const bool rowlane = lane < ns;
      const uint32_t slot = rowlane ? bv.unit_slots[(uint64_t)unit * ns_max + lane] : 0u;
      const uint32_t rsb = rowlane ? (uint32_t)(db.row_off[slot] * 4ull) : 0u;      // byte offset of the row in the postings array (< 4 GiB: host check)
      const uint32_t *pp = db.part2 + (uint64_t)slot * (np + 1u);                    // the row's line of the partition table (pp[0] = 0)
      s_slots[lane] = slot;
      s_c2[lane] = 0; s_cum[lane] = 0;
      // the unit's super-partitions: as many as keep ~ prm.clcap postings in each (the filter's false suspects grow with the cube of
      // what it holds: a short query scans its few hundred postings in ONE super-partition, a unit of heavy rows in many), in whole
      // partitions of the index's table; prm.W != 0 fixes the partitions per super-partition instead
      uint32_t SP = prm.W;
      if (SP == 0u) {
        const uint32_t P = (uint32_t)__builtin_amdgcn_readlane((int)r2_wave_incl_sum(rowlane ? pp[np] : 0u, lane), 63);
        const uint32_t want = (P + prm.clcap - 1u) / prm.clcap;            // super-partitions of equal size
        SP = (np + (want > 1u ? want : 1u) - 1u) / (want > 1u ? want : 1u);
        SP = SP < 1u ? 1u : SP;
      }
      // super-partition = partitions [pb, pe) of the index's table.  One that overflows the group-step list or the records is scanned
      // again as two halves (nothing of it was kept yet); a single partition that overflows defers the unit.  The rows' boundary at
      // the end of the NEXT super-partition is fetched one ahead (e_pref = pp[pref_p])
      uint32_t pb = 0, span = SP, segb = 0;
      uint32_t pref_p = SP < np ? SP : np;
      uint32_t e_pref = rowlane ? pp[pref_p] : 0u;
      while (pb < np) {
        R3_CLK(const unsigned long long q0 = clock64();)
        const uint32_t pe = pb + span < np ? pb + span : np;
        const uint32_t sege = pe == pref_p ? e_pref : (rowlane ? pp[pe] : 0u);
        if (pe < np) { pref_p = pe + span < np ? pe + span : np; e_pref = rowlane ? pp[pref_p] : 0u; }
#define R3_SPLIT_OR_DEFER(why) { if (pe - pb > 1u) { span = (pe - pb) >> 1; continue; } bad = true; R3_WHY(why); break; }
#define R3_NEXT_SUPER { pb = pe; segb = sege; continue; }
        const uint32_t n = sege - segb;                                    // postings of the segment (0 on lanes without a row)
        const uint32_t B = rsb + segb * 4u;                                // its first byte
        const uint32_t lead = (B >> 2) & 3u;                               // postings between the 16-byte boundary below B and B
        const uint32_t S = n ? (lead + n + 31u) >> 5 : 0u;                 // group-steps: 8 aligned quads = 32 posting places each
        const uint32_t incl = r2_wave_incl_sum(S, lane);
        const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (T == 0u) R3_NEXT_SUPER
        any_posting = true;
        if (T > R3_GSCAP) R3_SPLIT_OR_DEFER(UGS_CTR_T7)
        // zero the filter (4 KB: four 16-byte stores per lane), lay out the group-steps
        {
          uint4 z; z.x = z.y = z.z = z.w = 0;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
          for (uint32_t k = 0; k < R3_BW * 4u / 1024u; ++k) *(uint4 *)(smem + k * 1024u + lane * 16u) = z;
          const uint32_t st = incl - S, a0 = B & ~15u;
          for (uint32_t k = 0; r2_ballot(k < S) != 0ull; ++k) {
            if (k < S) {
              const uint32_t rem = lead + n - k * 32u;                     // places from this group-step's origin to the segment's end
              s_gs[st + k] = make_uint2(a0 + k * 128u, (k == 0u ? lead : 0u) | ((rem < 32u ? rem : 32u) << 2) | (lane << 8));
            }
          }
          if (lane < R3_GSPAD) s_gs[T + lane] = make_uint2(0u, 0u);           // what the ring reads ahead of the last chunk (<= 4 chunks + the last one's rest)
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        R3_CLK(const unsigned long long q1 = clock64();)
        const uint32_t nch = (T + 7u) >> 3;                                // chunks of eight group-steps
        uint32_t n_rec = 0;
        uint32_t lm[4];
        uint32_t S_t[4] = {0, 0, 0, 0}, S_old[4] = {0, 0, 0, 0}, S_m[4] = {0, 0, 0, 0}, S_row = 0;
        uint64_t S_vm[4] = {0, 0, 0, 0};
        bool pv = false;
        // a chunk's descriptor: byte offset of the lane's quad, first place | end place << 2 | row << 8 (0: nothing valid)
        auto fetch = [&](uint32_t step, uint32_t &voff, uint32_t &m) {
          const uint2 d = s_gs[step * 8u + grp];                          // (behind the list: 40 entries of nothing, see the layout)
          voff = d.x + lo16; m = d.y;
        };
        // the postings of the counted chunk that passed: pass 1 the suspects' targets, pass 2 (row, target) records
#define R3_EMIT(P2)                                                                                               \
        {                                                                                                         \
          uint64_t hm[4];                                                                                         \
          _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                           \
            hm[j] = S_vm[j] & r2_ballot((P2) ? (S_old[j] & S_m[j]) != 0u : (S_old[j] & S_m[j]) == S_m[j]);        \
          if ((hm[0] | hm[1] | hm[2] | hm[3]) != 0ull) {                                                          \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) if (hm[j] != 0ull) {      /* (a chunk has a few hits: one or two of its four posting places) */ \
              const uint32_t pos = n_rec + r2_mbcnt(hm[j]);                                                       \
              if (((hm[j] >> lane) & 1ull) && pos < R3_RCAP + 64u) s_rec[pos] = (P2) ? (S_t[j] | (S_row << 24)) : S_t[j]; \
              n_rec += (uint32_t)__popcll(hm[j]);                                                                 \
            }                                                                                                     \
          }                                                                                                       \
          pv = false;                                                                                             \
        }
        // one stage: wait for slot k, issue the chunk four ahead into it, emit the previous chunk's hits (their LDS round trip lies
        // behind the posting wait), count this one.  Places outside a segment take no part: their lanes are masked out of the LDS
        // instruction (fewer bank conflicts) and out of the hits (S_vm: the valid lanes as scalar masks)
#define R3_STAGE(k, P2)                                                                                           \
        {                                                                                                         \
          uint32_t Tq[4];                                                                                         \
          R3_CLK2(const unsigned long long w0 = clock64();)                                                       \
          r2_take<k>(Tq);                                                                                         \
          R3_CLK2(tw += clock64() - w0;)                                                                          \
          const uint32_t cm = lm[k];                                                                              \
          { uint32_t vo; fetch(step + 4u + (uint32_t)(k), vo, lm[k]); r2_issue<k>(vo, postings); }                \
          if (pv) R3_EMIT(P2)                                                                                     \
          const uint32_t first = cm & 3u, span = ((cm >> 2) & 63u) - first, rel = i0 - first;                     \
          S_row = cm >> 8;                                                                                        \
          _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            const uint32_t t = Tq[j];                                                                             \
            const bool valid = rel + (uint32_t)j < span;                  /* (unsigned: places below `first` wrap) */ \
            const uint32_t ad = (t >> 3) & ((R3_BW - 1u) << 2);                                                   \
            S_t[j] = t;                                                                                           \
            S_vm[j] = r2_ballot(valid);                                                                           \
            if (P2) {                                                                                             \
              S_m[j] = 1u << (t & 31u);                                                                           \
              if (valid) S_old[j] = *(lds32)(uintptr_t)ad;               /* (the filter lies at LDS address 0: no static LDS in this kernel) */                        \
            } else {                                                                                              \
              const uint32_t h2 = __umulhi(t & 0xffffffu, R3_HMUL);       /* (v_mul_hi_u32_u24) */                \
              S_m[j] = (1u << (t & 31u)) | (1u << (h2 & 31u)) R3_THIRD_BIT(h2);                                   \
              if (valid) S_old[j] = __hip_atomic_fetch_or((lds32)(uintptr_t)ad, S_m[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
            }                                                                                                     \
          }                                                                                                       \
          pv = true;                                                                                              \
        }
#define R3_PASS(P2)                                                                                               \
        {                                                                                                         \
          { uint32_t vo;                                                                                          \
            fetch(0u, vo, lm[0]); r2_issue<0>(vo, postings); fetch(1u, vo, lm[1]); r2_issue<1>(vo, postings);     \
            fetch(2u, vo, lm[2]); r2_issue<2>(vo, postings); fetch(3u, vo, lm[3]); r2_issue<3>(vo, postings); }   \
          for (uint32_t step = 0; ; step += 4u) {                            /* (the loop ends behind the last chunk: what the ring still holds are loads of nothing) */ \
            R3_STAGE(0, P2) if (step + 1u >= nch) break;                                                          \
            R3_STAGE(1, P2) if (step + 2u >= nch) break;                                                          \
            R3_STAGE(2, P2) if (step + 3u >= nch) break;                                                          \
            R3_STAGE(3, P2) if (step + 4u >= nch) break;                                                          \
          }                                                                                                       \
          /* every load of the ring has landed before anything issues into the same slots again */                \
          asm volatile("s_waitcnt vmcnt(0)" : : : "memory");                                                     \
          if (pv) R3_EMIT(P2)                                                                                     \
        }
        // ---- pass 1: the filter
        R3_PASS(false)
        R3_CLK(const unsigned long long q2 = clock64();)
        R3_STAT(UGS_CTR_T0, n_rec); R3_STAT(UGS_CTR_T2, T); R3_STAT(UGS_CTR_T3, 1); R3_STATMAX(UGS_CTR_T4, n_rec);
        if (n_rec > R3_RCAP) R3_SPLIT_OR_DEFER(UGS_CTR_T5)
        if (n_rec == 0u) { R3_CLK(tc[0] += q1 - q0; tc[1] += q2 - q1;) R3_NEXT_SUPER }       // no target of this super-partition is touched twice
        // ---- the suspects' exact bits (target mod 32 768) into the zeroed words
        {
          uint4 z; z.x = z.y = z.z = z.w = 0;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
          for (uint32_t k = 0; k < R3_BW * 4u / 1024u; ++k) *(uint4 *)(smem + k * 1024u + lane * 16u) = z;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          for (uint32_t i = lane; i < n_rec; i += 64u) { const uint32_t t = s_rec[i]; atomicOr(&s_bm[(t >> 5) & (R3_BW - 1u)], 1u << (t & 31u)); }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          n_rec = 0;
        }
        // ---- pass 2: every occurrence of the suspects
        R3_PASS(true)
#undef R3_PASS
#undef R3_STAGE
#undef R3_EMIT
        R3_CLK(const unsigned long long q3 = clock64();)
        R3_STAT(UGS_CTR_T1, n_rec);
        if (n_rec > R3_RCAP) R3_SPLIT_OR_DEFER(UGS_CTR_T6)
        // ---- group the records by target: a table of R3_TAB entries over the filter's words and the group-step list (both done with):
        // the target as the key, the rows of its records as a 64-bit mask - count = its bits, first-touch row = its lowest bit
        {
          const uint32_t nr = n_rec;
          uint32_t *t_key = s_bm;
          unsigned long long *t_mask = (unsigned long long *)(smem + R3_TAB * 4u);
          uint4 f; f.x = f.y = f.z = f.w = 0xffffffffu;
          uint4 z; z.x = z.y = z.z = z.w = 0;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          static_assert(R3_TAB == 512u && R3_TAB * 12u <= R3_BW * 4u + R3_GSCAP * 8u, "the table lies over the filter and the group-step list");
#pragma unroll
          for (uint32_t k = 0; k < 2u; ++k) *(uint4 *)(smem + k * 1024u + lane * 16u) = f;
#pragma unroll
          for (uint32_t k = 2u; k < 6u; ++k) *(uint4 *)(smem + k * 1024u + lane * 16u) = z;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          for (uint32_t e0 = 0; e0 < nr; e0 += 64u) {
            const uint32_t i = e0 + lane;
            if (i < nr) {
              const uint32_t rec = s_rec[i], t = rec & 0xffffffu;
              uint32_t h = (t ^ (t >> 9)) & (R3_TAB - 1u);
              for (;;) {
                const uint32_t old = atomicCAS(&t_key[h], 0xffffffffu, t);
                if (old == 0xffffffffu || old == t) break;
                h = (h + 1u) & (R3_TAB - 1u);
              }
              atomicOr(&t_mask[h], 1ull << (rec >> 24));
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          // a target's key comes from its record of the lowest row (one record per row); count-2 keys are pruned as in k_rank2g: dropped
          // when K count-2 keys of EARLIER super-partitions (smaller targets) with the same or a lower row are kept already
          uint32_t nk_new = nk;
          for (uint32_t e0 = 0; e0 < nr; e0 += 64u) {
            const uint32_t i = e0 + lane;
            const bool act = i < nr;
            const uint32_t rec = act ? s_rec[i] : 0u, t = rec & 0xffffffu, row = rec >> 24;
            uint32_t cnt = 0, mrow = 0xffffffffu;
            if (act) {
              uint32_t h = (t ^ (t >> 9)) & (R3_TAB - 1u);
              while (t_key[h] != t) h = (h + 1u) & (R3_TAB - 1u);
              const unsigned long long mk = t_mask[h];
              cnt = (uint32_t)__popcll(mk); mrow = (uint32_t)__builtin_ctzll(mk);
            }
            const bool keep = act && cnt >= 2u && row == mrow && (cnt >= 3u || s_cum[row & 63u] < K);
            const uint64_t key = ((uint64_t)(255u - cnt) << 32) | rec;
            const uint64_t m = r2_ballot(keep);
            uint32_t pos = nk_new + r2_mbcnt(m);
            pos = pos < kcap ? pos : kcap;
            if (keep) s_kl[pos] = key;
            if (keep && cnt == 2u) atomicAdd(&s_c2[row & 63u], 1u);
            nk_new += (uint32_t)__popcll(m);
          }
          nk = nk_new;
          if (nk > kcap) { bad = true; R3_WHY(UGS_CTR_T7); break; }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          s_cum[lane] = r2_wave_incl_sum(s_c2[lane], lane);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        R3_CLK(tc[0] += q1 - q0; tc[1] += q2 - q1; tc[2] += q3 - q2; tc[3] += clock64() - q3;)
        pb = pe; segb = sege;
      }
#undef R3_SPLIT_OR_DEFER
#undef R3_NEXT_SUPER
    }
    if (bad) {
      if (lane == 0) {
        const unsigned long long idx = atomicAdd(&bv.counters[UGS_CTR_DEFER], 1ull);
        bv.defer_list[idx] = unit;
      }
      continue;
    }
    ++n_done_local;
    R3_CLK(const unsigned long long q4 = clock64();)
    r2g_finish_unit(db, bv, postings, unit, lane, ns, K, nk, any_posting, s_kl, s_fpk, s_sel, s_slots);
    R3_CLK(tc[4] += clock64() - q4;)
  }
#ifdef R3_CLOCKS      // layout | pass 1 | suspects + pass 2 | grouping | end of the unit
  if (lane == 0) { atomicAdd(&bv.counters[UGS_CTR_T0], tc[0]); atomicAdd(&bv.counters[UGS_CTR_T1], tc[1]); atomicAdd(&bv.counters[UGS_CTR_T2], tc[2]); atomicAdd(&bv.counters[UGS_CTR_T3], tc[3]); atomicAdd(&bv.counters[UGS_CTR_T4], tc[4]); }
#endif
#ifdef R3_CLOCKS2
  if (lane == 0) atomicAdd(&bv.counters[UGS_CTR_T5], tw);
#endif
  if (lane == 0 && n_done_local) atomicAdd(&bv.counters[UGS_CTR_R2_DONE], n_done_local);
}

size_t ugs_rank3g_lds(uint32_t kcap)
{
  return (size_t)r3_fixed_bytes() + ((size_t)kcap + 2) * 8;                  // (the kernel's own carve)
}

int ugs_rank3g_blocks_per_cu(size_t lds)
{
  int n = 0;
  const void *fn = (const void *)k_rank3g;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 64, lds) != hipSuccess || n < 1) n = 1;
  return n;
}

int ugs_launch_rank3g(const UgsDbView &db, const UgsBatchView &b, const UgsRank2Params &prm, int grid, hipStream_t st)
{
  if (b.cand_key || prm.gather != 2u || (prm.W == 0u && prm.clcap == 0u) || prm.np > 63u) { ugs_set_error("filter ranking kernel: outside its envelope"); return UGS_E_ENVELOPE; }
  const size_t need = ugs_rank3g_lds(prm.kcap);
  if (prm.lds < need) { ugs_set_error("filter ranking kernel: %u bytes of LDS per wave, its carve needs %zu", prm.lds, need); return UGS_E_ENVELOPE; }
  const void *fn = (const void *)k_rank3g;
  HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)prm.lds));
  UgsDbView a0 = db; UgsBatchView a1 = b; UgsRank2Params a2 = prm;
  void *args[] = {&a0, &a1, &a2};
  if (ugs_kernel_log) ugs_before_launch("k_rank3g");
  HIPCHK(hipLaunchKernel(fn, dim3(grid), dim3(64), args, prm.lds, st));
  if (ugs_kernel_log) ugs_after_launch("k_rank3g", st);
  HIPCHK(hipGetLastError());
  return UGS_OK;
}
