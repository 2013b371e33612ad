// ugs_inbatch.hip - device side of cluster_fast beyond the search kernels:
//   * growing the centroid index in place of UDBData::AddSIToDB_CopyData / AddWord / GrowRow (udbbuild.cpp:74-128,286-291):
//     the new centroids' words are indexed on their own (ugs_build_index) and every row of the result is appended to the
//     matching row of the resident CSR index (new targets have the largest indexes, so rows stay ascending);
//   * k_inbatch: the word counts U[] (udbusortedsearcher.cpp:375-410 / udbusortedsearcherbig.cpp:82-100) of every query
//     of a batch against the EARLIER queries of the same batch - the sequences that may have become centroids since the
//     batch's frozen index was searched.  The host (ugs_cluster.cpp) merges them into each query's candidate walk.
#include "ugs_dev.h"
#include <cstdio>
#include <algorithm>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ugs_set_error("%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return UGS_E_HIP; } } while (0)

#define CL_POS_BITS 44
#define CL_CMAXV 4095u
#define CL_KEY_INF 0xffffffffffffffffull

// ---- index growth
__global__ void k_merge_row_off(const uint64_t *old_off, const uint64_t *delta_off, uint32_t slots, uint64_t *new_off)
{
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s <= slots) new_off[s] = old_off[s] + delta_off[s];
}

// one wavefront per index row: old row, then the delta row with the target base added
__global__ void k_merge_rows(const uint64_t *old_off, const uint32_t *old_post, const uint64_t *delta_off, const uint32_t *delta_post,
                             const uint64_t *new_off, uint32_t slots, uint32_t base_target, uint32_t *new_post)
{
  const uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (s >= slots) return;
  const uint64_t oa = old_off[s], ob = old_off[s + 1], da = delta_off[s], db = delta_off[s + 1], na = new_off[s];
  for (uint64_t k = lane; k < ob - oa; k += 64) new_post[na + k] = old_post[oa + k];
  const uint64_t nb = na + (ob - oa);
  for (uint64_t k = lane; k < db - da; k += 64) new_post[nb + k] = delta_post[da + k] + base_target;
}

__global__ void k_max_row2(const uint64_t *row_off, uint32_t slots, uint32_t *max_row)
{
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t v = 0;
  if (s < slots) v = (uint32_t)(row_off[s + 1] - row_off[s]);
  for (int o = 32; o > 0; o >>= 1) { uint32_t x = __shfl_down(v, o); v = x > v ? x : v; }
  if ((threadIdx.x & 63) == 0 && v) atomicMax(max_row, v);
}

// new_off[slots+1], new_post (capacity >= old + delta + 256) are caller-allocated; d_max_row is a device word
int ugs_index_merge(const uint64_t *old_off, const uint32_t *old_post, const uint64_t *delta_off, const uint32_t *delta_post,
                    uint32_t slots, uint32_t base_target, uint64_t *new_off, uint32_t *new_post, uint64_t n_total,
                    uint32_t *d_max_row, hipStream_t st)
{
  hipLaunchKernelGGL(k_merge_row_off, dim3((slots + 1 + 255) / 256), dim3(256), 0, st, old_off, delta_off, slots, new_off);
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL(k_merge_rows, dim3((unsigned)(((uint64_t)slots * 64 + 255) / 256)), dim3(256), 0, st, old_off, old_post, delta_off,
                     delta_post, new_off, slots, base_target, new_post);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemsetAsync(new_post + n_total, 0, 256 * sizeof(uint32_t), st));        // padding: rows are read in whole-wave units
  HIPCHK(hipMemsetAsync(d_max_row, 0, 4, st));
  hipLaunchKernelGGL(k_max_row2, dim3((slots + 255) / 256), dim3(256), 0, st, new_off, slots, d_max_row);
  HIPCHK(hipGetLastError());
  return UGS_OK;
}

// ---- in-batch word counts
template <int CB> struct CTbl {
  static constexpr uint32_t MASK = (1u << CB) - 1u;
  static __device__ __forceinline__ void inc(uint32_t *t, uint32_t x) { atomicAdd(&t[(x * CB) >> 5], 1u << ((x * CB) & 31)); }
  // count of x, cleared in the same operation
  static __device__ __forceinline__ uint32_t take(uint32_t *t, uint32_t x)
  {
    const uint32_t sh = (x * CB) & 31;
    return (atomicAnd(&t[(x * CB) >> 5], ~(MASK << sh)) >> sh) & MASK;
  }
};

// One wavefront per unit (query strand).  The unit's sampled index rows are those k_rank_setup chose for the search of the
// frozen index (same query words, same step).  Rows of the batch's own index hold the batch's queries as targets; only
// targets j < the unit's own query can be in the database when the reference searches that query.
//   pass 1: U[j]++ over the sampled rows (LDS table of CB-bit counters, one per batch sequence)
//   pass 2: rows in scan order; a posting's counter is read and cleared at once, so only the first touch of a target sees
//           its count; kept: (count, first position) sorts before the key at which the frozen walk ended (`tkey`) - nothing
//           behind it can be reached by the merged walk (ugs_cluster.cpp).
// Entries (j, count | row << 16) in scan order.  Two launch shapes (r6; before: a count-only launch, a host scan, a second full launch):
//   slot_cap > 0  every unit: its count into ent_n[unit] and its first slot_cap entries into ent[unit * slot_cap ..] - a unit has about
//                 one entry on average (C3: 5.3 M entries for 5 M reads), so this one launch settles nearly all units;
//   slot_cap == 0 only the units of `list` (the ones that had more): all their entries at ent_off[unit].
template <int CB>
__global__ __launch_bounds__(256) void k_inbatch(UgsBatchView bv, const uint64_t *brow_off, const uint32_t *bpost, uint32_t ns_max,
                                                 uint32_t small_path, uint32_t max_rej, uint32_t tbl_words, uint32_t *ent_n,
                                                 const uint32_t *ent_off, uint2 *ent, uint32_t slot_cap, const uint32_t *list, uint32_t n_list)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wpb = blockDim.x >> 6;      // (wave index in an SGPR)
  const size_t wave_bytes = ((size_t)tbl_words + 64) * 4 + (size_t)ns_max * 8;
  uint32_t *tbl = (uint32_t *)(smem + (size_t)wave * wave_bytes);
  uint32_t *ra = tbl + tbl_words + 64, *re = ra + ns_max;
  for (uint32_t k = lane; k < tbl_words + 64; k += 64) tbl[k] = 0;
  const uint32_t units = list ? n_list : bv.nq * bv.nstrand, K = bv.K;
  for (uint32_t ui = blockIdx.x * wpb + wave; ui < units; ui += gridDim.x * wpb) {
    const uint32_t unit = list ? list[ui] : ui;
    const uint32_t qi = unit / bv.nstrand;
    const uint32_t ns = bv.unit_ns[unit];
    uint32_t n_out = 0;
    if (qi != 0 && ns != 0) {
      // where the frozen walk ended: at an accept, at the max_rej-th reject, or never (list exhausted)
      const uint32_t wn = bv.walk_n[unit], hn = bv.hit_n[unit];
      uint64_t tkey = CL_KEY_INF;
      if (wn && (hn != 0 || wn - hn >= max_rej)) tkey = bv.cand_key[(uint64_t)unit * K + wn - 1];
      const uint32_t *slots = bv.unit_slots + (uint64_t)unit * ns_max;
      // pass 1
      for (uint32_t r0 = 0; r0 < ns; r0 += 64) {
        const uint32_t r = r0 + lane;
        if (r < ns) {
          const uint32_t slot = slots[r];
          const uint32_t a = (uint32_t)brow_off[slot], b = (uint32_t)brow_off[slot + 1];
          uint32_t lo = a, hi = b;                                  // first posting >= qi
          while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (bpost[mid] < qi) lo = mid + 1; else hi = mid; }
          ra[r] = a; re[r] = lo;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      for (uint32_t r = 0; r < ns; ++r) {
        const uint32_t a = ra[r], e = re[r];
        for (uint32_t k = a + lane; k < e; k += 64) CTbl<CB>::inc(tbl, bpost[k]);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // pass 2
      const uint32_t off = slot_cap ? unit * slot_cap : ent_off[unit];
      const uint32_t cap = slot_cap ? slot_cap : 0xffffffffu;
      for (uint32_t r = 0; r < ns; ++r) {
        const uint32_t a = ra[r], e = re[r];
        for (uint32_t k0 = a; k0 < e; k0 += 64) {
          const uint32_t k = k0 + lane;
          uint32_t j = 0, c = 0;
          if (k < e) { j = bpost[k]; c = CTbl<CB>::take(tbl, j); }
          const uint64_t pos = small_path ? (uint64_t)(0x80000000u | j) : (((uint64_t)r << 32) | (0x80000000u | j));
          const uint64_t key = ((uint64_t)(CL_CMAXV - c) << CL_POS_BITS) | pos;
          const bool keep = c != 0 && key < tkey;
          const uint64_t m = __ballot(keep);
          const uint32_t at = n_out + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
          if (keep && at < cap) ent[(uint64_t)off + at] = make_uint2(j, c | (r << 16));
          n_out += (uint32_t)__popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
    if (lane == 0 && slot_cap) ent_n[unit] = n_out;
  }
}

int ugs_launch_inbatch(const UgsBatchView &bv, const uint64_t *brow_off, const uint32_t *bpost, uint32_t ns_max, int small_path,
                       uint32_t max_rej, int num_cu, uint32_t *ent_n, const uint32_t *ent_off, uint2 *ent, hipStream_t st,
                       uint32_t slot_cap, const uint32_t *list, uint32_t n_list)
{
  const uint32_t units = list ? n_list : bv.nq * bv.nstrand;
  if (!units) return UGS_OK;
  if (!ent || (slot_cap == 0) != (list != nullptr)) { ugs_set_error("in-batch counts: slots for every unit, or a list of units with offsets"); return UGS_E_ARG; }
  const int bits = ns_max <= 15 ? 4 : (ns_max <= 255 ? 8 : 16);
  const uint32_t tbl_words = (uint32_t)(((uint64_t)bv.nq * bits + 31) / 32);
  const size_t wave_bytes = ((size_t)tbl_words + 64) * 4 + (size_t)ns_max * 8;
  int wpb = 4;
  while (wpb > 1 && wpb * wave_bytes > 160 * 1024) wpb >>= 1;
  if (wpb * wave_bytes > 160 * 1024) { ugs_set_error("in-batch counter table %zu bytes exceeds the LDS (batch too large)", wave_bytes); return UGS_E_ENVELOPE; }
  const size_t lds = wpb * wave_bytes;
  const void *fn = bits == 4 ? (const void *)k_inbatch<4> : bits == 8 ? (const void *)k_inbatch<8> : (const void *)k_inbatch<16>;
  HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = 1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * wpb, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((units + wpb - 1) / wpb, (uint64_t)num_cu * per_cu));
  UgsBatchView a0 = bv; uint32_t a3 = ns_max, a4 = (uint32_t)small_path, a5 = max_rej, a6 = tbl_words;
  void *args[] = {&a0, &brow_off, &bpost, &a3, &a4, &a5, &a6, &ent_n, &ent_off, &ent, &slot_cap, &list, &n_list};
  if (ugs_kernel_log) ugs_before_launch("k_inbatch");
  HIPCHK(hipLaunchKernel(fn, dim3(grid), dim3(64 * wpb), args, lds, st));
  if (ugs_kernel_log) ugs_after_launch("k_inbatch", st);
  HIPCHK(hipGetLastError());
  return UGS_OK;
}
