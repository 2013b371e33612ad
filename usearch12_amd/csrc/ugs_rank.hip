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
// partitions of 2^gshift targets; each WAVE owns a private LDS counter table (4/8/16-bit
// counters) for one partition at a time and walks the sampled rows' sub-rows twice:
//   pass 1  LDS atomic increment per posting                       (the reference's U[t]++)
//   pass 2  same postings in row order: read the final count, clear the counter (so the table
//           is clean for the next partition, no bulk zeroing), record the first-touch position
//           per count value and emit every target with count >= 2 exactly once.
// Row order inside a wave is program order, so no workgroup barrier is needed in the scan.
// The reference fully sorts ~38 k touched targets per query; its candidate loop can consume at
// most K = maxaccepts+maxrejects-1 of them, so only the K smallest keys
// (count desc, first-touch position asc) are selected, after applying the reference's
// "MinValue = prevMax/2" (and, on the small path, -bump) cut-offs exactly.
#include "ugs_dev.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ugs_set_error("%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return UGS_E_HIP; } } while (0)

#define POS_BITS 44
#define POS_MASK ((1ull << POS_BITS) - 1)
#define CMAXV 4095u
#define KEY_INF 0xffffffffffffffffull

__device__ __forceinline__ uint64_t make_key(uint32_t c, uint64_t pos) { return ((uint64_t)(CMAXV - c) << POS_BITS) | pos; }
__device__ __forceinline__ uint32_t key_count(uint64_t k) { return CMAXV - (uint32_t)(k >> POS_BITS); }
__device__ __forceinline__ uint32_t key_target(uint64_t k) { return (uint32_t)k; }

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

struct RankShared {
  uint32_t Nu, ns, step, emit_n, M, next_value, min_value, n_kept, n_sel, nev, fill_limit_valid;
  uint64_t fill_limit;       // small path: count-1 targets kept only below this position
  uint64_t last_key;
  uint64_t red[8];
  uint32_t wsum[8];
  uint32_t base;
};

// One pass over all partitions owned by this wave.  FILL=false: normal extraction.
// FILL=true: emit count-1 targets in position order, at most `need` per partition.
template <int BITS, bool FILL>
__device__ void scan_partitions(const UgsDbView &db, const uint32_t *s_slots, uint32_t ns, uint32_t *tbl,
                                unsigned long long *s_fp, RankShared *sh, uint64_t *ebuf, uint64_t ecap,
                                int wave, int wpb, int lane, bool small_path, uint32_t need, uint64_t fill_limit)
{
  const uint32_t np = db.np, gshift = db.gshift;
  const uint64_t *row_off = db.row_off;
  const uint32_t *part = db.part;
  const uint32_t *postings = db.postings;
  for (uint32_t p = wave; p < np; p += wpb) {
    const uint32_t base_t = p << gshift;
    // ---- pass 1: count
    for (uint32_t i0 = 0; i0 < ns; i0 += 64) {
      uint64_t a = 0, b = 0;
      if (i0 + lane < ns) {
        uint32_t slot = s_slots[i0 + lane];
        uint64_t rb = row_off[slot];
        const uint32_t *pp = part + (uint64_t)slot * (np + 1) + p;
        a = rb + pp[0]; b = rb + pp[1];
      }
      const uint32_t nrows = (ns - i0) < 64 ? (ns - i0) : 64;
      for (uint32_t r = 0; r < nrows; ++r) {
        const uint64_t ra = shfl64(a, r), rbb = shfl64(b, r);
        for (uint64_t k = ra + lane; k < rbb; k += 64) Tbl<BITS>::inc(tbl, postings[k] - base_t);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint32_t quota_used = 0;
    if (FILL && small_path) {
      // small path scans U[] in ascending target order (udbusortedsearcher.cpp:230-267): walk the
      // table itself, emit the first `need` count-1 targets of this partition, clear as we go
      constexpr uint32_t EPW = 32 / BITS;
      const uint32_t words = (1u << gshift) / EPW;
      for (uint32_t w0 = 0; w0 < words; w0 += 64) {
        const uint32_t wi = w0 + lane;
        uint32_t word = 0;
        if (wi < words) { word = tbl[wi]; if (word) tbl[wi] = 0; }
        uint32_t mine = 0;
        for (uint32_t e2 = 0; e2 < EPW; ++e2) {
          const uint32_t c = (word >> (e2 * BITS)) & Tbl<BITS>::MASK;
          const uint64_t t = (uint64_t)base_t + (uint64_t)wi * EPW + e2;
          if (c == 1 && t < fill_limit) ++mine;
        }
        uint32_t incl = mine;
        for (int o = 1; o < 64; o <<= 1) { uint32_t x = __shfl_up((int)incl, o); if (lane >= o) incl += x; }
        const uint32_t total = __builtin_amdgcn_readlane((int)incl, 63);
        if (total && quota_used < need) {
          const uint32_t take = (need - quota_used) < total ? (need - quota_used) : total;
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(&sh->emit_n, take);
          base = __builtin_amdgcn_readfirstlane(base);
          uint32_t r = incl - mine;
          for (uint32_t e2 = 0; e2 < EPW; ++e2) {
            const uint32_t c = (word >> (e2 * BITS)) & Tbl<BITS>::MASK;
            const uint64_t t = (uint64_t)base_t + (uint64_t)wi * EPW + e2;
            if (c == 1 && t < fill_limit) {
              if (r < take && (uint64_t)base + r < ecap) ebuf[base + r] = make_key(1, t);
              ++r;
            }
          }
        }
        quota_used += total;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      continue;
    }
    // ---- pass 2: extract in row order
    for (uint32_t i0 = 0; i0 < ns; i0 += 64) {
      uint64_t a = 0, b = 0;
      if (i0 + lane < ns) {
        uint32_t slot = s_slots[i0 + lane];
        uint64_t rb = row_off[slot];
        const uint32_t *pp = part + (uint64_t)slot * (np + 1) + p;
        a = rb + pp[0]; b = rb + pp[1];
      }
      const uint32_t nrows = (ns - i0) < 64 ? (ns - i0) : 64;
      for (uint32_t r = 0; r < nrows; ++r) {
        const uint64_t ra = shfl64(a, r), rbb = shfl64(b, r);
        const uint32_t i = i0 + r;
        for (uint64_t k0 = ra; k0 < rbb; k0 += 64) {
          const uint64_t k = k0 + lane;
          uint32_t t = 0, c = 0;
          if (k < rbb) {
            t = postings[k];
            c = Tbl<BITS>::get(tbl, t - base_t);
            if (c) Tbl<BITS>::clear(tbl, t - base_t);
          }
          const uint64_t pos = small_path ? (uint64_t)t : (((uint64_t)i << 32) | t);
          bool e;
          if (!FILL) {
            if (c) { if (pos < s_fp[c]) atomicMin(&s_fp[c], (unsigned long long)pos); }
            e = c >= 2;
          } else {
            e = (c == 1) && (pos < fill_limit);
          }
          const uint64_t m = __ballot(e);
          if (m) {
            const uint32_t n = __popcll(m);
            const uint32_t rank = __popcll(m & ((1ull << lane) - 1ull));
            uint32_t take = n;
            if (FILL) { take = (quota_used >= need) ? 0 : ((need - quota_used) < n ? (need - quota_used) : n); quota_used += n; }
            if (take) {
              uint32_t base = 0;
              if (lane == 0) base = atomicAdd(&sh->emit_n, take);
              base = __builtin_amdgcn_readfirstlane(base);
              if (e && rank < take && (uint64_t)base + rank < ecap) ebuf[base + rank] = make_key(c, pos);
            }
          }
          // the clear must land before the next row reads the table (same wave, in-order LDS)
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}

// Block-wide min of a 64-bit key (all threads get the result)
__device__ uint64_t block_min_u64(uint64_t v, RankShared *sh, int wave, int wpb, int lane)
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

__device__ uint32_t block_sum_u32(uint32_t v, RankShared *sh, int wave, int wpb, int lane)
{
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor((int)v, o);
  __syncthreads();
  if (lane == 0) sh->wsum[wave] = v;
  __syncthreads();
  uint32_t r = 0;
  for (int w = 0; w < wpb; ++w) r += sh->wsum[w];
  return r;
}

template <int BITS>
__global__ void k_rank(UgsDbView db, UgsBatchView bv, uint32_t ns_max, uint32_t tbl_words)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, wpb = nthr >> 6;
  const uint32_t maxq = (bv.max_qlen + 15u) & ~15u;
  // LDS carve (all offsets multiples of 16)
  size_t off = 0;
  RankShared *sh = (RankShared *)(smem + off); off += (sizeof(RankShared) + 15) & ~(size_t)15;
  unsigned long long *s_fp = (unsigned long long *)(smem + off); off += (((size_t)ns_max + 1) * 8 + 15) & ~(size_t)15;
  uint32_t *s_words = (uint32_t *)(smem + off); off += (size_t)maxq * 4;
  uint32_t *s_slots = (uint32_t *)(smem + off); off += (((size_t)ns_max) * 4 + 15) & ~(size_t)15;
  uint8_t *s_q = (uint8_t *)(smem + off); off += maxq;
  uint8_t *s_first = (uint8_t *)(smem + off); off += maxq;
  uint32_t *s_ev_c = (uint32_t *)(smem + off); off += (((size_t)ns_max + 1) * 4 + 15) & ~(size_t)15;     // -bump events
  uint32_t *s_ev_minu = (uint32_t *)(smem + off); off += (((size_t)ns_max + 1) * 4 + 15) & ~(size_t)15;
  uint32_t *tbl = (uint32_t *)(smem + off) + (size_t)wave * tbl_words;

  const UgsTables *tab = db.tab;
  const uint32_t units = bv.nq * bv.nstrand;
  const uint32_t K = bv.K;
  const int W = db.word_len;
  const bool small_path = !db.big;
  uint64_t *ebuf = bv.emit_buf + (uint64_t)blockIdx.x * bv.emit_cap;
  const uint64_t ecap = bv.emit_cap;

  // counter tables must start clean; pass 2 restores that invariant after every partition
  for (uint32_t k = lane; k < tbl_words; k += 64) tbl[k] = 0;

  for (uint32_t unit = blockIdx.x; unit < units; unit += gridDim.x) {
    const uint32_t qi = unit / bv.nstrand, strand = unit % bv.nstrand;
    const uint64_t qo = bv.qoffs[qi];
    const uint32_t L = (uint32_t)(bv.qoffs[qi + 1] - qo);
    __syncthreads();
    // ---- query letters (reverse-complemented for strand 1: seqinfo.cpp:292-323)
    for (uint32_t p = tid; p < L; p += nthr) {
      uint8_t c;
      if (strand == 0) c = bv.qseqs[qo + p];
      else c = tab->comp[bv.qseqs[qo + (L - 1 - p)]];
      s_q[p] = c;
    }
    if (tid == 0) { sh->emit_n = 0; sh->n_sel = 0; sh->last_key = 0; }
    for (uint32_t c = tid; c <= ns_max; c += nthr) s_fp[c] = KEY_INF;
    __syncthreads();
    // ---- UDB words per position (udbparams.cpp:540-555)
    for (uint32_t p = tid; p < L; p += nthr) {
      uint32_t w = UGS_BAD_WORD;
      if (p + W <= L) {
        uint32_t acc = 0; bool ok = true;
        for (int k = 0; k < W; ++k) { uint32_t l = tab->udb_letter[s_q[p + k]]; ok = ok && (l != 0xff); acc = acc * db.alpha + l; }
        if (ok) w = acc;
      }
      s_words[p] = w;
    }
    __syncthreads();
    // ---- first occurrences (udbsearcher.cpp:161-194 keeps first-occurrence order)
    uint32_t my_first = 0;
    for (uint32_t p = tid; p < L; p += nthr) {
      const uint32_t w = s_words[p];
      bool first = (w != UGS_BAD_WORD);
      for (uint32_t q = 0; first && q < p; ++q) first = (s_words[q] != w);
      s_first[p] = first ? 1 : 0;
      my_first += first ? 1u : 0u;
    }
    const uint32_t Nu = block_sum_u32(my_first, sh, wave, wpb, lane);
    uint32_t step = 1;
    if (!small_path) step = db.step_tab[Nu < db.step_n ? Nu : db.step_n - 1];
    const uint32_t ns_q = Nu == 0 ? 0 : (Nu + step - 1) / step;
    const uint32_t ns = ns_q <= ns_max ? ns_q : ns_max;
    if (ns_q > ns_max && tid == 0) atomicOr(&bv.counters[UGS_CTR_ERR], (unsigned long long)UGS_ERR_NS);
    // ---- ranks of unique words -> sampled slots (every step-th unique word)
    {
      uint32_t run = 0;   // running count of firsts before the current tile
      for (uint32_t p0 = 0; p0 < L; p0 += nthr) {
        const uint32_t p = p0 + tid;
        const bool f = p < L && s_first[p];
        const uint64_t m = __ballot(f);
        const uint32_t inw = __popcll(m & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) sh->wsum[wave] = __popcll(m);
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (int w2 = 0; w2 < wpb; ++w2) { if (w2 < wave) before += sh->wsum[w2]; total += sh->wsum[w2]; }
        if (f) {
          const uint32_t rank = run + before + inw;
          if (rank % step == 0 && rank / step < ns) s_slots[rank / step] = s_words[p];
        }
        run += total;
      }
    }
    __syncthreads();
    // algorithmic postings P(q)
    {
      unsigned long long psum = 0;
      for (uint32_t i = tid; i < ns; i += nthr) { uint32_t s = s_slots[i]; psum += db.row_off[s + 1] - db.row_off[s]; }
      for (int o = 32; o > 0; o >>= 1) psum += __shfl_xor((long long)psum, o);
      if (lane == 0 && psum) atomicAdd(&bv.counters[UGS_CTR_POSTINGS], psum);
    }
    // ---- the scan
    scan_partitions<BITS, false>(db, s_slots, ns, tbl, s_fp, sh, ebuf, ecap, wave, wpb, lane, small_path, 0, 0);
    __threadfence_block();
    __syncthreads();

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
        if (small_path && m && db.bump_pct != 0) {
          // strict prefix maxima in ascending-target order = counts c whose first position
          // precedes the first position of every larger count (udbusortedsearcher.cpp:230-267)
          unsigned long long sufmin = KEY_INF;
          uint32_t nev = 0;
          for (uint32_t c = m; c >= 1; --c) {
            const unsigned long long f = s_fp[c];
            if (f != KEY_INF && f < sufmin) { s_ev_c[nev++] = c; sufmin = f; }   // descending c
          }
          // replay -bump over the events in scan order (ascending c); record (position, MinU after)
          const double Bump = db.bump_pct / 100.0;
          uint32_t MinU = 1, MaxCount = 0;
          uint32_t *ev_minu = s_ev_minu;
          for (int e = (int)nev - 1; e >= 0; --e) {
            const uint32_t n = s_ev_c[e];
            const uint32_t NewMin = (uint32_t)(n * Bump);
            if (NewMin > MinU && NewMin < MaxCount) MinU = NewMin;
            MaxCount = n;
            ev_minu[e] = MinU;
            if (MinU > 1 && sh->fill_limit == KEY_INF) sh->fill_limit = s_fp[n];
          }
          sh->nev = nev;
        }
      }
    }
    __syncthreads();
    const uint32_t min_value = sh->min_value;
    const uint32_t nev = sh->nev;
    const uint32_t *ev_minu = s_ev_minu;

    // kept(entry): count >= MinValue and (small path) count >= MinU in force at its position
    auto kept = [&](uint64_t key) -> bool {
      const uint32_t c = key_count(key);
      if (c < min_value) return false;
      if (nev) {
        const uint64_t pos = key & POS_MASK;
        uint32_t minu = 1;
        // events are stored by descending count == descending position; MinU in force at pos
        // = MinU after the last event strictly before pos
        for (uint32_t e = 0; e < nev; ++e) if (s_fp[s_ev_c[e]] < pos) { minu = ev_minu[e]; break; }
        if (c < minu) return false;
      }
      return true;
    };

    for (int phase = 0; phase < 2; ++phase) {
      const uint32_t N = sh->emit_n < ecap ? sh->emit_n : (uint32_t)ecap;
      // select the K smallest kept keys by repeated block-min above the previous one
      uint32_t nsel = sh->n_sel;
      uint64_t last = sh->last_key;
      bool exhausted = false;
      while (nsel < K) {
        uint64_t best = KEY_INF;
        for (uint32_t k = tid; k < N; k += nthr) {
          const uint64_t key = ebuf[k];
          if ((nsel == 0 || key > last) && key < best && kept(key)) best = key;
        }
        best = block_min_u64(best, sh, wave, wpb, lane);
        if (best == KEY_INF) { exhausted = true; break; }
        if (tid == 0) {
          bv.cand[(uint64_t)unit * K + nsel] = key_target(best);
          bv.cand_cnt[(uint64_t)unit * K + nsel] = key_count(best);
        }
        last = best; ++nsel;
      }
      __syncthreads();
      if (tid == 0) { sh->n_sel = nsel; sh->last_key = last; }
      __syncthreads();
      if (phase == 1 || !exhausted || nsel >= K) break;
      // fewer than K candidates with count >= 2: append count-1 targets in scan order if the
      // cut-offs keep them (MinValue <= 1; small path additionally position < first MinU bump)
      if (min_value > 1 || sh->M == 0) break;
      const uint32_t need = K - nsel;
      const uint64_t fill_limit = sh->fill_limit;
      __syncthreads();
      scan_partitions<BITS, true>(db, s_slots, ns, tbl, s_fp, sh, ebuf, ecap, wave, wpb, lane, small_path, need, fill_limit);
      __threadfence_block();
      __syncthreads();
    }
    if (tid == 0) {
      bv.cand_n[unit] = sh->n_sel;
      if (sh->emit_n > ecap) atomicOr(&bv.counters[UGS_CTR_ERR], (unsigned long long)UGS_ERR_EMIT);
    }
  }
}

int ugs_launch_rank(const UgsDbView &db, const UgsBatchView &b, const UgsRankLaunch &L, hipStream_t st)
{
  const uint32_t tbl_words = (uint32_t)((((uint64_t)1 << db.gshift) * L.bits) / 32);
  dim3 grid(L.grid), block(64 * L.wpb);
  switch (L.bits) {
  case 4:
    HIPCHK(hipFuncSetAttribute((const void *)k_rank<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds));
    hipLaunchKernelGGL(k_rank<4>, grid, block, L.lds, st, db, b, L.ns_max, tbl_words); break;
  case 8:
    HIPCHK(hipFuncSetAttribute((const void *)k_rank<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds));
    hipLaunchKernelGGL(k_rank<8>, grid, block, L.lds, st, db, b, L.ns_max, tbl_words); break;
  default:
    HIPCHK(hipFuncSetAttribute((const void *)k_rank<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds));
    hipLaunchKernelGGL(k_rank<16>, grid, block, L.lds, st, db, b, L.ns_max, tbl_words); break;
  }
  HIPCHK(hipGetLastError());
  return UGS_OK;
}
