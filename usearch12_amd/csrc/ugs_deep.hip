// ugs_deep.hip - deep candidate walks: the COMPLETE sorted candidate list of a unit, gfx950.
//
// The reference's candidate loop has no depth limit: Terminator::Terminate (terminator.cpp:22-31,64-100) ends a walk after
// -maxaccepts accepts or -maxrejects rejects, 0 = never, and the loop then runs to the end of the sorted target list
// (udbusortedsearcherbig.cpp:113-134, udbusortedsearcher.cpp:122-152).  The ranking kernels (ugs_rank.hip, ugs_rank2.hip) keep the
// UGS_KMAX = 64 best candidates of a unit, which is all that the default and every setting with max_accepts + max_rejects - 1 <= 64
// can visit.  A search with a deeper or an open walk (UGS_A_DEEP) runs as usual first; the few units whose walk used up a full list
// without meeting a limit are parked by k_align, and for those this file makes the whole list:
//
//   k_deep      one workgroup per parked unit, four walks over the unit's index rows (the rows k_rank_setup chose):
//               1. U[t] = rows that hold target t, R[t] = the first of them (global scratch of the workgroup, one word each per target)
//               2. fp[c] = first position of a target with count c - then MaxValue, NextValue, MinValue = NextValue / 2
//                  (countsort.cpp:13-24,114-126) and, on the small path, the -bump events (udbusortedsearcher.cpp:230-267):
//                  exactly the numbers k_rank derives from its emitted keys
//               3. every touched target once (from its first row): key = (count, first-touch position), kept by the same rule as
//                  k_rank's (count >= MinValue, count >= the MinU in force at its position, small path: not refused by a pair
//                  filter) - counted (mode 0) or written (mode 1)
//               4. U[] back to zero
//   ugs_deep_sort  rocPRIM segmented radix sort of the units' keys: ascending key order IS the reference's candidate order
//
// k_align then continues each parked walk through its list from candidate UGS_KMAX on (ugs_align.hip, continuation pass).
// Nothing here is on the path of a default search; it is sized for correctness at any depth, not for speed.
#include "ugs_rank_keys.h"
#include <cstring>
#include <algorithm>
#include <rocprim/rocprim.hpp>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ugs_set_error("%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return UGS_E_HIP; } } while (0)
#define DEEP_INF 0xffffffffffffffffull

__global__ __launch_bounds__(256) void k_deep(UgsDbView db, UgsBatchView bv, UgsDeepArgs a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const uint32_t ns_max = a.ns_max;
  unsigned long long *s_fp = (unsigned long long *)smem;                              // [ns_max + 1] first position per count value
  uint32_t *s_ev_c = (uint32_t *)(s_fp + ns_max + 1);                                   // [ns_max + 1] -bump events: count
  uint32_t *s_ev_minu = s_ev_c + ns_max + 1;                                            // ... MinU after the event
  uint32_t *s_slots = s_ev_minu + ns_max + 1;                                           // [ns_max] the unit's index rows
  __shared__ uint32_t s_M, s_minv, s_nev, s_cnt;
  uint32_t *U = a.U + (uint64_t)blockIdx.x * a.stride, *R = a.R + (uint64_t)blockIdx.x * a.stride;
  const bool small = !db.big;
  for (uint32_t i = blockIdx.x; i < a.n_units; i += gridDim.x) {
    const uint32_t unit = a.units[i];
    const uint32_t ns = bv.unit_ns[unit];
    __syncthreads();
    for (uint32_t r = tid; r < ns; r += nthr) s_slots[r] = bv.unit_slots[(uint64_t)unit * ns_max + r];
    for (uint32_t c = tid; c <= ns_max; c += nthr) s_fp[c] = DEEP_INF;
    if (tid == 0) { s_M = 0; s_minv = 0; s_nev = 0; s_cnt = 0; }
    __syncthreads();
    // ---- 1. counts and first rows (a row holds a target once: the first touch of a target is alone in its row's pass)
    for (uint32_t r = 0; r < ns; ++r) {
      const uint64_t ra = db.row_off[s_slots[r]], rb = db.row_off[s_slots[r] + 1];
      for (uint64_t k = ra + tid; k < rb; k += nthr) { const uint32_t t = db.postings[k]; if (atomicAdd(&U[t], 1u) == 0u) R[t] = r; }
      __syncthreads();
    }
    // ---- 2. first position per count value
    for (uint32_t r = 0; r < ns; ++r) {
      const uint64_t ra = db.row_off[s_slots[r]], rb = db.row_off[s_slots[r] + 1];
      for (uint64_t k = ra + tid; k < rb; k += nthr) {
        const uint32_t t = db.postings[k];
        if (R[t] == r) { const uint64_t pos = small ? (uint64_t)t : (((uint64_t)r << 32) | t); atomicMin(&s_fp[U[t]], (unsigned long long)pos); }
      }
    }
    __syncthreads();
    if (tid == 0) {
      // NextValue = the running maximum just before the maximum last rose, in scan order = max{c < M : fp[c] < fp[M]}
      uint32_t m = 0;
      for (uint32_t c = 1; c <= ns; ++c) if (s_fp[c] != DEEP_INF) m = c;
      uint32_t nv = 0;
      if (m) for (uint32_t c = 1; c < m; ++c) if (s_fp[c] < s_fp[m]) nv = c;
      s_M = m; s_minv = nv / 2;
      uint32_t nev = 0;
      if (small && m && db.bump_pct != 0) {
        // strict prefix maxima in ascending-target order, then -bump replayed over them (udbusortedsearcher.cpp:230-267)
        unsigned long long sufmin = DEEP_INF;
        for (uint32_t c = m; c >= 1; --c) { const unsigned long long f = s_fp[c]; if (f != DEEP_INF && f < sufmin) { s_ev_c[nev++] = c; sufmin = f; } }
        const double Bump = db.bump_pct / 100.0;
        uint32_t MinU = 1, MaxCount = 0;
        for (int e = (int)nev - 1; e >= 0; --e) {
          const uint32_t n = s_ev_c[e];
          const uint32_t NewMin = (uint32_t)(n * Bump);
          if (NewMin > MinU && NewMin < MaxCount) MinU = NewMin;
          MaxCount = n;
          s_ev_minu[e] = MinU;
        }
      }
      s_nev = nev;
    }
    __syncthreads();
    const uint32_t min_value = s_minv, nev = s_nev;
    PairQ pq;
    pq.mask = small ? (db.pair_mask & ~(uint32_t)UGS_P_SELFID) : 0u;
    if (pq.mask) {
      const uint32_t qi = unit / bv.nstrand;
      pq.ql = (uint32_t)(bv.qoffs[qi + 1] - bv.qoffs[qi]);
      pq.qkey = bv.q_key ? bv.q_key[qi] : 0u; pq.qsize = bv.q_size ? bv.q_size[qi] : 0xffffffffu;
      pq.min_sizeratio = db.min_sizeratio; pq.minqt = db.minqt; pq.maxqt = db.maxqt; pq.minsl = db.minsl; pq.maxsl = db.maxsl;
      pq.offs = db.offs; pq.t_key = db.t_key; pq.t_size = db.t_size;
    }
    auto kept = [&](uint32_t c, uint64_t pos, uint32_t t) -> bool {
      if (c < min_value) return false;
      if (nev) {
        uint32_t minu = 1;      // (events by descending count = descending position: MinU after the last event strictly before pos)
        for (uint32_t e = 0; e < nev; ++e) if (s_fp[s_ev_c[e]] < pos) { minu = s_ev_minu[e]; break; }
        if (c < minu) return false;
      }
      if (pq.mask && pair_reject(pq, t)) return false;
      return true;
    };
    // ---- 3. the kept keys: counted, or written behind the unit's offset
    uint64_t *out = a.mode ? a.keys + a.key_off[i] : nullptr;
    for (uint32_t r = 0; r < ns; ++r) {
      const uint64_t ra = db.row_off[s_slots[r]], rb = db.row_off[s_slots[r] + 1];
      for (uint64_t k = ra + tid; k < rb; k += nthr) {
        const uint32_t t = db.postings[k];
        if (R[t] != r) continue;
        const uint32_t c = U[t];
        const uint64_t pos = small ? (uint64_t)t : (((uint64_t)r << 32) | t);
        if (!kept(c, pos, t)) continue;
        const uint32_t slot = atomicAdd(&s_cnt, 1u);
        if (out) out[slot] = make_key(c, pos);
      }
    }
    __syncthreads();
    if (tid == 0 && !a.mode) a.key_n[i] = s_cnt;
    // ---- 4. the counters of the touched targets back to zero
    for (uint32_t r = 0; r < ns; ++r) {
      const uint64_t ra = db.row_off[s_slots[r]], rb = db.row_off[s_slots[r] + 1];
      for (uint64_t k = ra + tid; k < rb; k += nthr) U[db.postings[k]] = 0u;
    }
    __syncthreads();
  }
}

int ugs_launch_deep(const UgsDbView &db, const UgsBatchView &b, const UgsDeepArgs &a, int grid, hipStream_t st)
{
  const size_t lds = ((size_t)a.ns_max + 1) * 8 + 2 * ((size_t)a.ns_max + 1) * 4 + (size_t)a.ns_max * 4 + 64;
  HIPCHK(hipFuncSetAttribute((const void *)k_deep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_deep, dim3(grid), dim3(256), lds, st, db, b, a);
  HIPCHK(hipGetLastError());
  return UGS_OK;
}

// every unit's keys ascending (segment i = [d_off[i], d_off[i + 1])); *d_tmp grows on demand and stays with the caller
int ugs_deep_sort(uint64_t *d_keys, uint64_t *d_sorted, uint64_t total, uint32_t segments, const uint64_t *d_off, void **d_tmp, size_t *tmp_bytes, hipStream_t st)
{
  if (!total || !segments) return UGS_OK;
  if (total > 0xffffffffull) { ugs_set_error("deep walk: %llu keys in one chunk", (unsigned long long)total); return UGS_E_ENVELOPE; }
  size_t need = 0;
  HIPCHK(rocprim::segmented_radix_sort_keys(nullptr, need, d_keys, d_sorted, (unsigned int)total, segments, d_off, d_off + 1, 0, 64, st));
  if (need > *tmp_bytes) {
    if (*d_tmp) HIPCHK(ugs_free(*d_tmp));
    *d_tmp = nullptr; *tmp_bytes = 0;
    HIPCHK(ugs_malloc(d_tmp, need + 256));
    *tmp_bytes = need + 256;
  }
  HIPCHK(rocprim::segmented_radix_sort_keys(*d_tmp, need, d_keys, d_sorted, (unsigned int)total, segments, d_off, d_off + 1, 0, 64, st));
  return UGS_OK;
}
