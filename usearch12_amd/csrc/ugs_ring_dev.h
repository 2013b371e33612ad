// ugs_ring_dev.h - device helpers shared by the bitmap ranking kernels (ugs_rank2.hip: k_rank2, k_rank2g; ugs_rank3.hip: k_rank3g):
// ballots, DPP prefix sums, the ring of posting loads in accumulator registers, and the end of a sparse-index unit (cut-offs,
// all-pairs ranking of the kept keys, count-1 fill).  Internal, gfx950 only.
#pragma once
#include "ugs_dev.h"
#include "ugs_rank2.h"

typedef uint32_t __attribute__((address_space(3))) *lds32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// the lanes' predicate as a 64-bit mask: ONE v_cmp into a scalar pair (HIP's __ballot(int) compares a 0 / 1 value made by v_cndmask
// against zero again wherever the predicate is a conjunction)
__device__ __forceinline__ uint64_t r2_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ uint32_t r2_mbcnt(uint64_t m)
{
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ uint64_t r2_readlane64(uint64_t v, uint32_t l)
{
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l);
  return ((uint64_t)hi << 32) | lo;
}
// inclusive prefix sum inside each row of 16 lanes (DPP row shifts)
__device__ __forceinline__ uint32_t r2_row16_incl_sum(uint32_t v)
{
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
  return v;
}

// The ring's posting loads land in ACCUMULATOR registers a[4k : 4k+3] (slot k).  They are asm statements the compiler does not
// count, so that their waits can be counted by hand (vmcnt(D - 1)) - and the accumulator file is where such a load is safe:
// the compiler never touches those registers (this kernel has no MFMA and spills nothing: tests/test_isa.py pins both), whereas a
// VGPR destination may be copied or re-used by the register allocator between the load and its wait (seen at a loop
// back-edge).  r2_take<K> waits for slot K and DEFINES four fresh VGPR values from it.
// H (half-width ring, k_rank2's 16-bit postings): slot k = a[2k : 2k+1], an 8-byte load per lane; r2_take defines t[0], t[1] only.
template <int K, bool H = false> __device__ __forceinline__ void r2_issue(uint32_t voff, const uint32_t *src)
{
  static_assert(K >= 0 && K < 4, "four ring slots");
  if constexpr (H) {
    if constexpr (K == 0) asm volatile("global_load_dwordx2 a[0:1], %0, %1" : : "v"(voff), "s"(src) : "memory", "a0", "a1");
    if constexpr (K == 1) asm volatile("global_load_dwordx2 a[2:3], %0, %1" : : "v"(voff), "s"(src) : "memory", "a2", "a3");
    if constexpr (K == 2) asm volatile("global_load_dwordx2 a[4:5], %0, %1" : : "v"(voff), "s"(src) : "memory", "a4", "a5");
    if constexpr (K == 3) asm volatile("global_load_dwordx2 a[6:7], %0, %1" : : "v"(voff), "s"(src) : "memory", "a6", "a7");
    return;
  }
  if constexpr (K == 0) asm volatile("global_load_dwordx4 a[0:3], %0, %1" : : "v"(voff), "s"(src) : "memory", "a0", "a1", "a2", "a3");
  if constexpr (K == 1) asm volatile("global_load_dwordx4 a[4:7], %0, %1" : : "v"(voff), "s"(src) : "memory", "a4", "a5", "a6", "a7");
  if constexpr (K == 2) asm volatile("global_load_dwordx4 a[8:11], %0, %1" : : "v"(voff), "s"(src) : "memory", "a8", "a9", "a10", "a11");
  if constexpr (K == 3) asm volatile("global_load_dwordx4 a[12:15], %0, %1" : : "v"(voff), "s"(src) : "memory", "a12", "a13", "a14", "a15");
}
template <int K, bool H = false> __device__ __forceinline__ void r2_take(uint32_t (&t)[4])
{
  if constexpr (H) {
    if constexpr (K == 0) asm volatile("s_waitcnt vmcnt(3)\n\tv_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1" : "=v"(t[0]), "=v"(t[1]) : : "memory");
    if constexpr (K == 1) asm volatile("s_waitcnt vmcnt(3)\n\tv_accvgpr_read_b32 %0, a2\n\tv_accvgpr_read_b32 %1, a3" : "=v"(t[0]), "=v"(t[1]) : : "memory");
    if constexpr (K == 2) asm volatile("s_waitcnt vmcnt(3)\n\tv_accvgpr_read_b32 %0, a4\n\tv_accvgpr_read_b32 %1, a5" : "=v"(t[0]), "=v"(t[1]) : : "memory");
    if constexpr (K == 3) asm volatile("s_waitcnt vmcnt(3)\n\tv_accvgpr_read_b32 %0, a6\n\tv_accvgpr_read_b32 %1, a7" : "=v"(t[0]), "=v"(t[1]) : : "memory");
    return;
  }
  if constexpr (K == 0) asm volatile("s_waitcnt vmcnt(3)\n\tv_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3" : "=v"(t[0]), "=v"(t[1]), "=v"(t[2]), "=v"(t[3]) : : "memory");
  if constexpr (K == 1) asm volatile("s_waitcnt vmcnt(3)\n\tv_accvgpr_read_b32 %0, a4\n\tv_accvgpr_read_b32 %1, a5\n\tv_accvgpr_read_b32 %2, a6\n\tv_accvgpr_read_b32 %3, a7" : "=v"(t[0]), "=v"(t[1]), "=v"(t[2]), "=v"(t[3]) : : "memory");
  if constexpr (K == 2) asm volatile("s_waitcnt vmcnt(3)\n\tv_accvgpr_read_b32 %0, a8\n\tv_accvgpr_read_b32 %1, a9\n\tv_accvgpr_read_b32 %2, a10\n\tv_accvgpr_read_b32 %3, a11" : "=v"(t[0]), "=v"(t[1]), "=v"(t[2]), "=v"(t[3]) : : "memory");
  if constexpr (K == 3) asm volatile("s_waitcnt vmcnt(3)\n\tv_accvgpr_read_b32 %0, a12\n\tv_accvgpr_read_b32 %1, a13\n\tv_accvgpr_read_b32 %2, a14\n\tv_accvgpr_read_b32 %3, a15" : "=v"(t[0]), "=v"(t[1]), "=v"(t[2]), "=v"(t[3]) : : "memory");
}

__device__ __forceinline__ uint32_t r2_wave_incl_sum(uint32_t v, uint32_t lane)
{
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
  const uint32_t t0 = __builtin_amdgcn_readlane((int)v, 15), t1 = __builtin_amdgcn_readlane((int)v, 31), t2 = __builtin_amdgcn_readlane((int)v, 47);
  const uint32_t row = lane >> 4;
  return v + (row >= 1 ? t0 : 0u) + (row >= 2 ? t1 : 0u) + (row >= 3 ? t2 : 0u);
}
__device__ __forceinline__ uint32_t r2_wave_incl_max(uint32_t v)          // (unsigned, identity 0)
{
  uint32_t x;
  x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); v = x > v ? x : v;
  x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); v = x > v ? x : v;
  x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); v = x > v ? x : v;
  x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); v = x > v ? x : v;
  x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); v = x > v ? x : v;      // row_bcast:15 -> rows 1, 3
  x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); v = x > v ? x : v;      // row_bcast:31 -> rows 2, 3
  return v;
}
#define R2G_KEY_INF 0xffffffffffffffffull
#define R2G_MAXROWS 63u

// ---- the end of a sparse-index unit (k_rank2g, k_rank3g): cut-offs from the smallest key of every count value (countsort.cpp:13-24,
// 114-126), all-pairs ranking of the nk kept 64-bit keys ((255 - count) << 32 | first row << 24 | target), the K smallest with count >=
// MinValue written out, and - where fewer than K exist and MinValue <= 1 - the count-1 fill in first-touch order (the rows walked in
// ascending order against the selected list, as k_rank's big_path_fill).  s_kl holds nk keys and room for two sentinels.
__device__ __forceinline__ void r2g_finish_unit(const UgsDbView &db, const UgsBatchView &bv, const uint32_t *postings, uint32_t unit, uint32_t lane,
                                                uint32_t ns, uint32_t K, uint32_t nk, bool any_posting, uint64_t *s_kl, uint64_t *s_fpk, uint32_t *s_sel,
                                                const uint32_t *s_slots)
{
  // ---- cut-offs from the smallest key of every count value (countsort.cpp:13-24,114-126)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  s_fpk[lane] = R2G_KEY_INF;
  if (lane < 2u) s_kl[nk + lane] = R2G_KEY_INF;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (uint32_t i = lane; i < nk; i += 64u) { const uint64_t key = s_kl[i]; atomicMin((unsigned long long *)&s_fpk[(255u - (uint32_t)(key >> 32)) & 63u], (unsigned long long)key); }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  uint32_t M = 0, nv = 0;
  {
    const uint64_t f = (lane >= 2u) ? s_fpk[lane] : R2G_KEY_INF;            // lane c looks at count c
    const uint64_t vm = r2_ballot(f != R2G_KEY_INF);
    if (vm) {
      M = 63u - (uint32_t)__builtin_clzll(vm);
      const uint32_t fM = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)f, (int)M);
      const uint64_t lmk = r2_ballot(f != R2G_KEY_INF && lane < M && (uint32_t)f < fM);
      nv = lmk ? 63u - (uint32_t)__builtin_clzll(lmk) : 0u;
    } else M = any_posting ? 1u : 0u;
  }
  const uint32_t min_value = nv / 2u;
  const uint32_t cmin = min_value > 2u ? min_value : 2u;
  const uint64_t limit = (uint64_t)(256u - cmin) << 32;                   // keys below it have count >= cmin
  uint32_t nsel = 0;
  {
    uint32_t nelig = 0;
    for (uint32_t e0 = 0; e0 < nk; e0 += 64u) {
      const uint32_t i = e0 + lane;
      const uint64_t key = i < nk ? s_kl[i] : R2G_KEY_INF;
      const bool elig = key < limit;
      nelig += (uint32_t)__popcll(r2_ballot(elig));
      uint32_t rank = 0;
      for (uint32_t j = 0; j < nk; j += 2u) { rank += (s_kl[j] < key) + (s_kl[j + 1u] < key); }
      if (elig && rank < K) {
        const uint32_t tg = (uint32_t)key & 0xffffffu;
        bv.cand[(uint64_t)unit * K + rank] = tg;
        bv.cand_cnt[(uint64_t)unit * K + rank] = 255u - (uint32_t)(key >> 32);
        s_sel[rank] = tg;
      }
    }
    nsel = nelig < K ? nelig : K;
  }
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
        if (e && filled + rank < K) { bv.cand[(uint64_t)unit * K + filled + rank] = t; bv.cand_cnt[(uint64_t)unit * K + filled + rank] = 1u; }
        const uint32_t n = (uint32_t)__popcll(m);
        filled = filled + n < K ? filled + n : K;
      }
    }
    nsel = filled;
  }
  if (lane == 0) bv.cand_n[unit] = nsel;
}
