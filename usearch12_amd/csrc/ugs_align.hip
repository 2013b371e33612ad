// ugs_align.hip - the global aligner + accept/terminate replay on gfx950.
// One WAVE per (query, strand) walks that unit's ranked candidates serially with the
// reference's early-termination rule; inside a pair the 64 lanes share the work.
//
// Replaces (reference, /root/reference/src):
//   candidate loop + Terminator          udbusortedsearcherbig.cpp:113-134, terminator.cpp:64-100
//   HSPFinder::SetA/SetB/SeqToWords      hspfinder.cpp:226-331
//   HSPFinder::UngappedBlast             ungappedblast.cpp:8-211   (lanes = target seed positions;
//                                         the serial "BPos = Bhi+1" skip is replayed by ballot)
//   IsGlobalHSP / IsStaggered / Chain    hspfinder.cpp:594-636, hsp.h:102-126, chainer.cpp:352-500
//   GetGlobalHSPs / gate                 getglobalhsps.cpp:9-61, globalalignmem.cpp:161-180
//   GetHole / AlignHSPMem / AlnParams::Init   globalalignmem.cpp:25-112, alnparams.cpp:100-152
//   ViterbiFastBandMem (+MainDiag)       viterbifastbandmem.cpp:12-253  (row sweep: lanes = band
//                                         columns, M/D by shuffle from the previous row, the
//                                         in-row I recurrence by a wavefront max-plus prefix scan)
//   TraceBackBitMem                      tracebackbitmem.cpp:8-73
//   AlignResult::FillLo / GetGapOpenCount     arscorer.cpp:201-296,554-569
//   Accepter::IsAcceptLo (-id)           accepter.cpp:27-94
// Scores are int32 in units of 0.5 (all reference scores are exact half-integers);
// MINUS_INFINITY is a saturating sentinel so that -inf + x == -inf and -inf >= -inf hold as in fp32.
#include "ugs_dev.h"
#include <algorithm>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ugs_set_error("%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return UGS_E_HIP; } } while (0)

#define NEG (-(1 << 29))
#define NEGT (-(1 << 28))
#define TB_DM 1
#define TB_IM 2
#define TB_MD 4
#define TB_MI 8
#define LTB 1024               // per-wave LDS traceback bytes (holes up to ~25x36 cells)
#define LRUNS 32               // runs kept in LDS per wave before spilling to HBM scratch

#ifndef UGS_TB_RUNS
#define UGS_TB_RUNS 1        // viterbi_hole's traceback takes a run of match columns in one trip
#endif
#ifndef UGS_ALIGN_CLOCKS
#define UGS_ALIGN_CLOCKS 0
#endif
__device__ __forceinline__ int sat_add(int x, int c) { return x <= NEGT ? NEG : x + c; }
__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
// the lane index made on the spot (two instructions).  A phase that takes its lane from here instead of the kernel's `lane` does not
// keep the loop-invariant values the compiler derives from it (lane masks, lane * stride addresses) alive across the whole unit loop:
// those were the VGPRs k_align spilled - stored once per kernel, loaded back once per unit (r5)
__device__ __forceinline__ int fresh_lane() { int l; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l)); return l; }

struct HSPd { uint32_t Loi, Loj, Len; int32_t Score2; };
// a value every lane of the wave holds alike (an LDS word read at a wave-uniform address, a flag made of such words), moved to the scalar
// file: what follows from it - loop bounds, branch conditions, the walk's counters - is then scalar too instead of one VGPR each
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ HSPd uni(const HSPd &h) { HSPd r; r.Loi = uni(h.Loi); r.Loj = uni(h.Loj); r.Len = uni(h.Len); r.Score2 = (int32_t)uni((uint32_t)h.Score2); return r; }

struct WaveState {            // per-wave LDS control block (written by lane 0, read by all after a wave fence)
  uint32_t nhsp, nchain, nruns, cur_op, cur_len, rt_n, overflow, pad;
};

struct Pen { int OpenA, OpenB, ExtA, ExtB, LOpenA, LOpenB, LExtA, LExtB, ROpenA, ROpenB, RExtA, RExtB; };

struct WaveCtx {
  // LDS
  uint8_t *A, *B;             // class codes of query / target letters (identity tests)
  uint8_t *As, *Bs;           // score codes: nt 0..3 = A,C,G,T/U, 4 = anything else; aa = letter 0..25 (31 other)
  uint32_t *A2, *Ai, *B2, *Bi; // nt: the score codes packed 2 bits per letter (16 letters per word) and, in the same layout, one bit per
                              // letter that is not A/C/G/T/U (bit 2k of its word); two zero words in front, three behind
  uint16_t *wstart;           // per HSP word: first index in qsort (12 bits) | min(count,8) << 12 (0 = absent), or null.  With more than
                              // 1024 words (aa: 8000) the table is per BUCKET (word >> bsh): first index | entries << 12 (<= 15; a query
                              // with a fuller bucket sorts and searches the general way: use_tab false); qsort is then ordered by
                              // (bucket, position) and a look-up filters the bucket's entries by word
  uint32_t bsh; bool use_tab;
  uint32_t *seeds; uint32_t seed_cap; uint32_t union_words;   // seed list of the current pair: bpos << 16 | apos, in reference order
  bool nt;
  uint32_t *qsort;            // sorted (hsp word << 16 | pos) of the query
  int32_t *Mrow, *Drow;       // Mrow[-1] valid
  HSPd *hsps;
  uint32_t *chain;            // indexes into hsps in chain order
  uint32_t *csc;              // chainer scratch
  WaveState *ws;
  const uint8_t *s_cls; const int8_t *s_sub2; const uint64_t *s_match; const uint8_t *s_hl;
  const uint16_t *xlut;       // the extension table (extend_nt_lut) or null
  // global scratch
  uint8_t *tb; uint32_t *runs; uint32_t runs_cap;
  uint8_t *lds_tb; uint32_t *lds_runs; uint32_t *lds_rt;   // LDS fast copies: small traceback matrices, first LRUNS runs
  uint32_t LA, LB, nwA, nA2, hsp_cap, nwords;
  bool a_inv, b_inv;          // nt: the query / the target holds a letter that is not A/C/G/T/U (wave-uniform)
  int lane;
};

__device__ __forceinline__ int score2(const WaveCtx &c, uint8_t a, uint8_t b) { return c.s_sub2[((a & 31) << 5) | (b & 31)]; }
// score of two SCORE codes.  nt: pure arithmetic (setnucmx.cpp:11-99: ACGTU match/mismatch by letter,
// anything else scores 0); aa: BLOSUM62 row lookup in LDS
template <bool NT>
__device__ __forceinline__ int sscore(const WaveCtx &c, int m2, int mm2, uint32_t a, uint32_t b)
{
  if (NT) return (a < 4 && b < 4) ? (a == b ? m2 : mm2) : 0;
  return c.s_sub2[(a << 5) | b];
}
__device__ __forceinline__ bool ident(const WaveCtx &c, uint8_t a, uint8_t b) { return (c.s_match[a] >> b) & 1ull; }

// wave-wide inclusive prefix sum with DPP row shifts (no LDS crossbar round trips)
__device__ __forceinline__ uint32_t wave_incl_sum_u32(uint32_t v)
{
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
  const uint32_t t0 = __builtin_amdgcn_readlane((int)v, 15), t1 = __builtin_amdgcn_readlane((int)v, 31), t2 = __builtin_amdgcn_readlane((int)v, 47);
  const uint32_t row = (uint32_t)fresh_lane() >> 4;
  return v + (row >= 1 ? t0 : 0u) + (row >= 2 ? t1 : 0u) + (row >= 3 ? t2 : 0u);
}
// LDS-only hand-off inside one wave: the LDS unit executes a wave's operations in program order,
// so only the compiler must be kept from reordering
__device__ __forceinline__ void lds_sync() { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); }

__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }

// hspfinder.cpp:594-636
__device__ __forceinline__ bool is_global_hsp(uint32_t ALo, uint32_t BLo, uint32_t LA, uint32_t LB)
{
  if (LA <= LB) {
    uint32_t MaxGap = LA / 4 + 1;
    if (ALo > BLo && ALo - BLo > MaxGap) return false;
    uint32_t AR = LA - ALo, BR = LB - BLo;
    if (AR > BR && AR - BR > MaxGap) return false;
  } else {
    uint32_t MaxGap = LB / 4 + 1;
    if (BLo > ALo && BLo - ALo > MaxGap) return false;
    uint32_t AR = LA - ALo, BR = LB - BLo;
    if (BR > AR && BR - AR > MaxGap) return false;
  }
  return true;
}

// nt HSP-finder word of w letters at `pos`, read from the packed letters: the same letters as the reference's rolling word
// (hspfinder.cpp:226-270, invalid letter -> 0) in little-endian digit order - query table and target look-ups both use it
__device__ __forceinline__ uint32_t nt_word(const uint32_t *w2, uint32_t pos, int w)
{
  const uint32_t k = pos >> 4;
  return __builtin_amdgcn_alignbit(w2[k + 1], w2[k], (pos & 15u) * 2u) & ((1u << (2 * w)) - 1u);
}

// Query side of HSPFinder::SetA: words for every position (invalid letter -> 0), sorted by
// (word, pos) so a word's first MaxReps positions are contiguous and ascending.
__device__ __forceinline__ void build_query_words(WaveCtx &c, int w, int alpha)
{
  const int lane = fresh_lane();
  const uint32_t LA = c.LA;
  c.nwA = LA >= (uint32_t)w ? LA - w + 1 : 0;
  uint32_t n2 = 64; while (n2 < c.nwA) n2 <<= 1;
  c.nA2 = n2;
  c.use_tab = c.wstart != nullptr;
  bool counting = c.wstart && (uint64_t)c.nwA * 3 / 2 + 16 <= c.union_words;
  if (counting) {
    // ---- counting sort by word (<= 1024 words; or by bucket of words), stable in position: O(L^2/64) equal-word ranks with
    // 128-bit LDS reads instead of a 36-stage bitonic network
    uint32_t *tw = c.seeds;                               // word per position (union region, free here)
    uint16_t *tr = (uint16_t *)(c.seeds + ((c.nwA + 3) & ~3u));       // rank among earlier equal words (buckets)
    const uint32_t bsh = c.bsh, ntab = ((c.nwords - 1u) >> bsh) + 1u, nwA = c.nwA;
    for (uint32_t k = lane; k < (ntab + 1) / 2; k += 64) ((uint32_t *)c.wstart)[k] = 0;
    for (uint32_t p = lane; p < ((nwA + 3) & ~3u); p += 64) {
      uint32_t word = 0xffffffffu;
      if (p < nwA) { if (c.nt) word = nt_word(c.A2, p, w); else { word = 0; for (int k = 0; k < w; ++k) word = word * alpha + c.s_hl[c.A[p + k] & 31]; } }
      tw[p] = word;
    }
    lds_sync();
    for (uint32_t p = lane; p < nwA; p += 64) {
      const uint32_t bk = tw[p] >> bsh;
      atomicAdd(&((uint32_t *)c.wstart)[bk >> 1], 1u << ((bk & 1u) * 16));   // 16-bit counters, two per LDS word
    }
    lds_sync();
    if (bsh) {                                             // a bucket's entry count has four bits in the table
      bool over = false;
      for (uint32_t k = lane; k < ntab; k += 64) over = over || c.wstart[k] > 15;
      if (__ballot(over)) { counting = false; c.use_tab = false; }
    }
  }
  if (counting) {
    uint32_t *tw = c.seeds;
    uint16_t *tr = (uint16_t *)(c.seeds + ((c.nwA + 3) & ~3u));
    const uint32_t bsh = c.bsh, ntab = ((c.nwords - 1u) >> bsh) + 1u, nwA = c.nwA;
    // only the positions of words that occur more than once need a rank (about a fifth of a random query): they are listed
    // densely first, so that the O(L) rank scans fill whole wavefronts
    uint32_t *dup = c.qsort;                               // (free until the final scatter)
    // ... and their words (buckets) densely behind the rank array when the union region has the room: a position's rank is then
    // counted among the EARLIER LISTED positions only (a quarter of the scan over all earlier positions)
    uint32_t *dw = c.seeds + ((((nwA + 3) & ~3u) * 3u / 2u + 3u) & ~3u);
    const bool dense = (uint64_t)((((nwA + 3) & ~3u) * 3u / 2u + 3u) & ~3u) + nwA + 4u <= c.union_words;
    uint32_t ndup = 0;
    for (uint32_t p0 = 0; p0 < nwA; p0 += 64) {
      const uint32_t p = p0 + lane;
      const uint32_t wd = tw[p < nwA ? p : 0] >> bsh;
      const bool need = p < nwA && c.wstart[wd] > 1;
      if (p < nwA) tr[p] = 0;
      const uint64_t m = __ballot(need);
      if (need) { const uint32_t k = ndup + __popcll(m & ((1ull << lane) - 1ull)); dup[k] = p; if (dense) dw[k] = wd; }
      ndup += (uint32_t)__popcll(m);
    }
    lds_sync();
    if (dense) {
      for (uint32_t i = lane; i < ndup; i += 64) {
        const uint32_t wd = dw[i];
        const uint4 *v4 = (const uint4 *)dw;
        uint32_t rank = 0;
        const uint32_t nq4 = i >> 2;
        for (uint32_t q4 = 0; q4 < nq4; ++q4) { const uint4 x = v4[q4]; rank += (x.x == wd) + (x.y == wd) + (x.z == wd) + (x.w == wd); }
        for (uint32_t q = nq4 << 2; q < i; ++q) rank += dw[q] == wd;
        tr[dup[i]] = (uint16_t)rank;
      }
    } else
    for (uint32_t i = lane; i < ndup; i += 64) {
      const uint32_t p = dup[i], wd = tw[p] >> bsh;
      const uint4 *v4 = (const uint4 *)tw;
      uint32_t rank = 0;
      const uint32_t nq4 = p >> 2;
      for (uint32_t q4 = 0; q4 < nq4; ++q4) { const uint4 x = v4[q4]; rank += ((x.x >> bsh) == wd) + ((x.y >> bsh) == wd) + ((x.z >> bsh) == wd) + ((x.w >> bsh) == wd); }
      for (uint32_t q = nq4 << 2; q < p; ++q) rank += (tw[q] >> bsh) == wd;
      tr[p] = (uint16_t)rank;
    }
    lds_sync();
    {   // exclusive prefix sum of the counts: each lane owns a contiguous block of words
      const uint32_t per = (ntab + 63) / 64;
      const uint32_t ncap = bsh ? 15u : (uint32_t)UGS_MAXREPS;
      uint32_t sum = 0;
      for (uint32_t k = 0; k < per; ++k) { const uint32_t wi = lane * per + k; if (wi < ntab) sum += c.wstart[wi]; }
      const uint32_t incl = wave_incl_sum_u32(sum);
      uint32_t run = incl - sum;
      for (uint32_t k = 0; k < per; ++k) {
        const uint32_t wi = lane * per + k;
        if (wi < ntab) { const uint32_t n = c.wstart[wi]; c.wstart[wi] = (uint16_t)(n ? (run | ((n < ncap ? n : ncap) << 12)) : 0u); run += n; }
      }
    }
    lds_sync();
    for (uint32_t p = lane; p < nwA; p += 64) {
      const uint32_t wd = tw[p];
      c.qsort[(c.wstart[wd >> bsh] & 0xfffu) + tr[p]] = (wd << 16) | p;
    }
    lds_sync();
    return;
  }
  if (c.bsh) c.use_tab = false;                            // (bucket tables are only built by the counting sort)
  for (uint32_t p = lane; p < n2; p += 64) {
    uint32_t key = 0xffffffffu;
    if (p < c.nwA) {
      uint32_t word = 0;
      if (c.nt) word = nt_word(c.A2, p, w); else for (int k = 0; k < w; ++k) word = word * alpha + c.s_hl[c.A[p + k] & 31];
      key = (word << 16) | p;
    }
    c.qsort[p] = key;
  }
  lds_sync();
  for (uint32_t k = 2; k <= n2; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = lane; i < n2; i += 64) {
        uint32_t l = i ^ j;
        if (l > i) {
          uint32_t x = c.qsort[i], y = c.qsort[l];
          bool up = ((i & k) == 0);
          if ((x > y) == up) { c.qsort[i] = y; c.qsort[l] = x; }
        }
      }
      lds_sync();
    }
  if (c.use_tab) {
    // direct word -> (first sorted index | min(count, MaxReps) << 16) table: replaces a binary
    // search per target position; count 0 = word absent from the query
    const uint32_t nwords = c.nwords;
    for (uint32_t k = lane; k < (nwords + 1) / 2; k += 64) ((uint32_t *)c.wstart)[k] = 0;
    lds_sync();
    for (uint32_t i = lane; i < c.nwA; i += 64) {
      const uint32_t wd = c.qsort[i] >> 16;
      if (i == 0 || (c.qsort[i - 1] >> 16) != wd) {
        uint32_t n = 1;
        while (n < UGS_MAXREPS && i + n < c.nwA && (c.qsort[i + n] >> 16) == wd) ++n;
        c.wstart[wd] = (uint16_t)(i | (n << 12));
      }
    }
    lds_sync();
  }
}

// 8 consecutive score codes starting at byte offset o of an LDS byte array (aligned dword reads +
// v_alignbyte); the arrays carry 16 bytes of padding on both sides
__device__ __forceinline__ uint64_t load8(const uint8_t *arr, int o)
{
  const uint32_t *w = (const uint32_t *)(arr + (o & ~3));
  const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
  const uint32_t sh = (uint32_t)o & 3u;
  const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, sh);
  const uint32_t hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
  return ((uint64_t)hi << 32) | lo;
}
// 0x80 in every byte of x that is non-zero
__device__ __forceinline__ uint64_t nzbytes(uint64_t x)
{
  return (((x & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | x) & 0x8080808080808080ull;
}

// nt score codes (bytes, 0..3 = A,C,G,T/U, 4 = other) -> 2 bits per letter + "other" bits in the same layout
__device__ __forceinline__ void pack_codes(const uint8_t *codes, uint32_t L, uint32_t *w2, uint32_t *wi)
{
  const int lane = fresh_lane();
  const uint32_t nw = (L + 15) >> 4;
  for (uint32_t k = lane; k < nw + 3; k += 64) {
    uint32_t v = 0, iv = 0;
    if (k < nw) {
      const uint4 d4 = *(const uint4 *)(codes + 16 * k);             // (the code arrays are 16-byte aligned and padded)
      const uint32_t d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t lo = d[q] & 0x03030303u, hi = (d[q] >> 2) & 0x01010101u;
        v |= ((lo | (lo >> 6) | (lo >> 12) | (lo >> 18)) & 0xffu) << (8 * q);
        iv |= ((hi | (hi >> 6) | (hi >> 12) | (hi >> 18)) & 0x55u) << (8 * q);
      }
    }
    w2[k] = v; wi[k] = iv;
  }
  if (lane < 2) { w2[-1 - lane] = 0; wi[-1 - lane] = 0; }
}
// 32 letters starting at letter `pos` (may be negative down to -32) of a packed array, 2 bits each
__device__ __forceinline__ uint64_t read32l(const uint32_t *w, int pos)
{
  const int k = pos >> 4;
  const uint32_t sh = ((uint32_t)pos & 15u) * 2u;
  const uint32_t w0 = w[k], w1 = w[k + 1], w2 = w[k + 2];
  return ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, sh) << 32) | __builtin_amdgcn_alignbit(w1, w0, sh);
}

// The nt extension on packed letters: 32 letter pairs per LDS round trip; inside a block the walk goes from one
// non-matching pair to the next (a run of matches is consumed at once: the score rises through it, so its best is its end
// and the x-drop test cannot fire inside it).  Same results as the byte-wise walk of ungappedblast.cpp:91-178.
template <bool INV>          // INV = false: neither sequence holds a non-ACGTU letter, the "other letter" planes are not read
__device__ __forceinline__ void extend_nt_packed(const uint32_t *A2, const uint32_t *Ai, const uint32_t *B2, const uint32_t *Bi, int m2, int mm2, int X, uint32_t LA, uint32_t LB,
                                                 uint32_t &a1, uint32_t &b1, uint32_t &a2, uint32_t &b2, int &score, int &best,
                                                 uint32_t &bestb1, uint32_t &bestb2)
{
  const uint64_t EVEN = 0x5555555555555555ull;
  {
    uint32_t rem = (LB - 1 - b2) < (LA - 1 - a2) ? (LB - 1 - b2) : (LA - 1 - a2);
    bool stop = false;
    while (rem && !stop) {
      const uint32_t n = rem < 32 ? rem : 32;
      const uint64_t x = read32l(A2, (int)a2 + 1) ^ read32l(B2, (int)b2 + 1);
      const uint64_t inv = INV ? (read32l(Ai, (int)a2 + 1) | read32l(Bi, (int)b2 + 1)) : 0ull;
      uint64_t bad = ((x | (x >> 1)) | inv) & EVEN;
      if (n < 32) bad |= 1ull << (2 * n);                          // sentinel behind the last pair of the block
      uint32_t pos = 0;
      for (;;) {
        const uint64_t b = bad >> (2 * pos);
        const uint32_t run = b ? (uint32_t)(__ffsll((long long)b) - 1) >> 1 : 32u - pos;
        if (run) { score += (int)run * m2; pos += run; if (score > best) { best = score; bestb2 = b2 + pos; } }
        if (pos >= n) break;
        score += (INV && ((inv >> (2 * pos)) & 1ull)) ? 0 : mm2;   // a pair with a non-ACGT letter scores 0 (setnucmx.cpp)
        ++pos;
        if (score > best) { best = score; bestb2 = b2 + pos; }
        else if (best - score > X) { stop = true; break; }
        if (pos >= n) break;
      }
      a2 += n; b2 += n; rem -= n;
    }
  }
  score = best;
  {
    uint32_t rem = b1 < a1 ? b1 : a1;
    bool stop = false;
    while (rem && !stop) {
      const uint32_t n = rem < 32 ? rem : 32;
      const uint64_t x = read32l(A2, (int)a1 - 32) ^ read32l(B2, (int)b1 - 32);     // pair 31 = position -1
      const uint64_t inv = INV ? (read32l(Ai, (int)a1 - 32) | read32l(Bi, (int)b1 - 32)) : 0ull;
      uint64_t bad = ((x | (x >> 1)) | inv) & EVEN;
      if (n < 32) bad |= 1ull << (2 * (31 - n));                   // sentinel in front of the first pair of the block
      uint32_t pos = 0;                                            // pairs consumed, from the top
      for (;;) {
        const uint64_t b = bad << (2 * pos);
        const uint32_t run = b ? (uint32_t)__clzll((long long)b) >> 1 : 32u - pos;
        if (run) { score += (int)run * m2; pos += run; if (score > best) { best = score; bestb1 = b1 - pos; } }
        if (pos >= n) break;
        score += (INV && ((inv >> (2 * (31 - pos))) & 1ull)) ? 0 : mm2;
        ++pos;
        if (score > best) { best = score; bestb1 = b1 - pos; }
        else if (best - score > X) { stop = true; break; }
        if (pos >= n) break;
      }
      a1 -= n; b1 -= n; rem -= n;
    }
  }
}

// The same extension (no letter outside A/C/G/T/U on either side) driven by a TABLE: the serial rule of ungappedblast.cpp:91-178
//   score += s; if (score > best) { best = score; pos = here; } else if (best - score > X) stop;
// depends on the past only through the deficit d = best - score (0 .. X), so four letter pairs at a time are one look-up of
// xlut[d / 2][mismatch bits of the four pairs] = stop << 15 | step of the last new best (1..4, 0 none) << 12 | rise of the best / 2 << 6 |
// new deficit / 2, built per workgroup by simulating the rule letter by letter (k_align).  Needs even match / mismatch scores and an
// even X <= 32 half-units (the reference's defaults: 2, -4, 32); k_align falls back to extend_nt_packed otherwise (c.xlut == null).
// A random seed stops after ~ 15 letters: four or five look-ups per direction instead of a dozen trips of the mismatch-to-mismatch
// loop - the seeds of a pair without a relative are where k_align's time goes.
__device__ __forceinline__ uint32_t compact16(uint32_t y)          // bits 0, 2, .. 30 of y -> bits 0 .. 15
{
  y = (y | (y >> 1)) & 0x33333333u;
  y = (y | (y >> 2)) & 0x0f0f0f0fu;
  y = (y | (y >> 4)) & 0x00ff00ffu;
  return (y | (y >> 8)) & 0xffffu;
}
__device__ __forceinline__ uint32_t read16l(const uint32_t *w, int pos)    // 16 letters starting at letter `pos` (>= -32)
{
  const int k = pos >> 4;
  return __builtin_amdgcn_alignbit(w[k + 1], w[k], ((uint32_t)pos & 15u) * 2u);
}
__device__ __forceinline__ void extend_nt_lut(const uint32_t *A2, const uint32_t *B2, const uint16_t *xlut, uint32_t LA, uint32_t LB,
                                              uint32_t &a1, uint32_t &b1, uint32_t &a2, uint32_t &b2, int &best,
                                              uint32_t &bestb1, uint32_t &bestb2)
{
  {
    uint32_t rem = (LB - 1 - b2) < (LA - 1 - a2) ? (LB - 1 - b2) : (LA - 1 - a2);
    uint32_t d = 0;
    bool stop = false;
    while (rem && !stop) {
      const uint32_t n = rem < 16 ? rem : 16;
      const uint32_t x = read16l(A2, (int)a2 + 1) ^ read16l(B2, (int)b2 + 1);
      uint32_t m = compact16((x | (x >> 1)) & 0x55555555u);
      if (n < 16) m |= 0xffffu << n;                                // behind the end: mismatches (they can stop the walk, never raise the best)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (!stop && (uint32_t)(4 * q) < n) {
          const uint32_t e = xlut[(d << 4) | ((m >> (4 * q)) & 15u)];
          const uint32_t g = (e >> 6) & 63u;
          if (g) { best += 2 * (int)g; bestb2 = b2 + 4u * q + ((e >> 12) & 7u); }
          d = e & 63u; stop = (e >> 15) != 0u;
        }
      }
      a2 += n; b2 += n; rem -= n;
    }
  }
  {
    uint32_t rem = b1 < a1 ? b1 : a1;
    uint32_t d = 0;
    bool stop = false;
    while (rem && !stop) {
      const uint32_t n = rem < 16 ? rem : 16;
      const uint32_t x = read16l(A2, (int)a1 - 16) ^ read16l(B2, (int)b1 - 16);        // letter 15 = position -1
      uint32_t m = __builtin_bitreverse32(compact16((x | (x >> 1)) & 0x55555555u)) >> 16;   // bit j = the pair at position -1 - j
      if (n < 16) m |= 0xffffu << n;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (!stop && (uint32_t)(4 * q) < n) {
          const uint32_t e = xlut[(d << 4) | ((m >> (4 * q)) & 15u)];
          const uint32_t g = (e >> 6) & 63u;
          if (g) { best += 2 * (int)g; bestb1 = b1 - 4u * q - ((e >> 12) & 7u); }
          d = e & 63u; stop = (e >> 15) != 0u;
        }
      }
      a1 -= n; b1 -= n; rem -= n;
    }
  }
}

// wave-wide inclusive prefix maximum (identity INT_MIN)
__device__ __forceinline__ int coop_incl_max(int v)
{
  const int NEGI = (int)0x80000000;
  int x;
  x = __builtin_amdgcn_update_dpp(NEGI, v, 0x111, 0xf, 0xf, false); v = x > v ? x : v;      // row_shr:1
  x = __builtin_amdgcn_update_dpp(NEGI, v, 0x112, 0xf, 0xf, false); v = x > v ? x : v;      // row_shr:2
  x = __builtin_amdgcn_update_dpp(NEGI, v, 0x114, 0xf, 0xf, false); v = x > v ? x : v;      // row_shr:4
  x = __builtin_amdgcn_update_dpp(NEGI, v, 0x118, 0xf, 0xf, false); v = x > v ? x : v;      // row_shr:8
  x = __builtin_amdgcn_update_dpp(NEGI, v, 0x142, 0xa, 0xf, false); v = x > v ? x : v;      // row_bcast:15 -> rows 1, 3
  x = __builtin_amdgcn_update_dpp(NEGI, v, 0x143, 0xc, 0xf, false); v = x > v ? x : v;      // row_bcast:31 -> rows 2, 3
  return v;
}

// The same seed extension (ungappedblast.cpp:62-180, byte score codes) by the WHOLE wave: one letter pair per lane and step.  The
// serial rule "score += s; if (score > best) best = score, pos = here; else if (best - score > X) stop" in prefix form: with P_i the
// running score and M_i the running maximum (the earlier best included), the walk stops at the first i with M_i - P_i > X (when P_i
// exceeds the earlier maximum the difference is 0), the best is M at the last consumed position and its position the first that
// attains it.  Used for the first seed of a round whose diagonal holds many of the round's seeds (a homologous pair: all those lanes
// would walk the same hundreds of letters, eight (aa) or a run of matches (nt) at a time); same outputs as extend_seed, wave-uniform.
template <bool NT>
__device__ __forceinline__ bool extend_seed_coop(const WaveCtx &c, const UgsDbView &db, int m2, int mm2, uint32_t apos, uint32_t bpos, uint32_t MinLength,
                                                 uint32_t &oAlo, uint32_t &oBlo, uint32_t &oLen, int &oBest)
{
  const int w = db.hsp_w, X = db.xdrop2;
  const uint32_t LA = c.LA, LB = c.LB, lane = (uint32_t)c.lane;
  const bool inv_any = NT && (c.a_inv || c.b_inv);
  int score = 0;
  if (NT) {
    const uint32_t ninv = inv_any ? (uint32_t)__popc((nt_word(c.Ai, apos, w) | nt_word(c.Bi, bpos, w))) : 0u;      // (as extend_seed)
    score = ((int)w - (int)ninv) * m2;
  } else
    for (int k = 0; k < w; ++k) score += (int)c.s_sub2[((uint32_t)c.As[apos + k] << 5) | c.Bs[bpos + k]];
  int best = score;
  uint32_t b2 = bpos + w - 1, a2 = apos + w - 1, bestb2 = b2;
  uint32_t a1 = apos, b1 = bpos, bestb1 = b1;
#pragma unroll
  for (int dir = 0; dir < 2; ++dir) {
    uint32_t rem = dir == 0 ? ((LB - 1 - b2) < (LA - 1 - a2) ? (LB - 1 - b2) : (LA - 1 - a2)) : (b1 < a1 ? b1 : a1);
    if (dir == 1) score = best;
    while (rem) {
      const uint32_t n = rem < 64u ? rem : 64u;
      const bool valid = lane < n;
      int sc = 0;
      if (valid) {
        const uint32_t pa = dir == 0 ? a2 + 1 + lane : a1 - 1 - lane, pb = dir == 0 ? b2 + 1 + lane : b1 - 1 - lane;
        if (NT) {
          const uint32_t la = (c.A2[pa >> 4] >> ((pa & 15u) * 2u)) & 3u, lb = (c.B2[pb >> 4] >> ((pb & 15u) * 2u)) & 3u;
          sc = la == lb ? m2 : mm2;
          if (inv_any && (((c.Ai[pa >> 4] >> ((pa & 15u) * 2u)) | (c.Bi[pb >> 4] >> ((pb & 15u) * 2u))) & 1u)) sc = 0;   // a pair with a non-ACGT letter scores 0 (setnucmx.cpp)
        } else
          sc = (int)c.s_sub2[((uint32_t)c.As[pa] << 5) | c.Bs[pb]];
      }
      const int P = score + (int)wave_incl_sum_u32((uint32_t)sc);
      int M = coop_incl_max(valid ? P : (int)0x80000000);
      M = M > best ? M : best;
      const uint64_t sm = __ballot(valid && M - P > X);
      const uint32_t lim = sm ? (uint32_t)__ffsll((long long)sm) - 1u : n;        // positions [0, lim) are consumed
      if (lim) {
        const int nb = rl(M, (int)lim - 1);
        if (nb > best) {
          const uint32_t j = (uint32_t)__ffsll((long long)__ballot(lane < lim && P == nb)) - 1u;
          if (dir == 0) bestb2 = b2 + 1 + j; else bestb1 = b1 - 1 - j;
          best = nb;
        }
      }
      if (sm) break;
      score = rl(P, (int)n - 1);
      if (dir == 0) { a2 += n; b2 += n; } else { a1 -= n; b1 -= n; }
      rem -= n;
    }
  }
  const uint32_t Blo = bestb1, Bhi = bestb2, Len = Bhi - Blo + 1;
  const uint32_t Alo = apos - (bpos - bestb1);
  oAlo = Alo; oBlo = Blo; oLen = Len; oBest = best;
  return Len >= MinLength && best >= db.minscore2 && is_global_hsp(Alo, Blo, LA, LB);
}

// The amino-acid extension (ungappedblast.cpp:91-178) on score-code bytes: 8 letter pairs per LDS round trip, consumed strictly in order
__device__ __forceinline__ void extend_aa_bytes(const int8_t *sub2, const uint8_t *As, const uint8_t *Bs, int X, uint32_t LA, uint32_t LB,
                                                uint32_t &a1, uint32_t &b1, uint32_t &a2, uint32_t &b2, int &score, int &best,
                                                uint32_t &bestb1, uint32_t &bestb2)
{
  {
    uint32_t rem = (LB - 1 - b2) < (LA - 1 - a2) ? (LB - 1 - b2) : (LA - 1 - a2);
    bool stop = false;
    while (rem && !stop) {
      uint32_t av[8], bv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { const uint32_t o = (uint32_t)(k + 1) <= rem ? (uint32_t)(k + 1) : rem; av[k] = As[a2 + o]; bv[k] = Bs[b2 + o]; }
      const uint32_t n = rem < 8 ? rem : 8;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if ((uint32_t)k < n && !stop) {
          score += (int)sub2[(av[k] << 5) | bv[k]];
          if (score > best) { best = score; bestb2 = b2 + k + 1; }
          else if (best - score > X) stop = true;
        }
      a2 += n; b2 += n; rem -= n;
    }
  }
  score = best;
  {
    uint32_t rem = b1 < a1 ? b1 : a1;
    bool stop = false;
    while (rem && !stop) {
      uint32_t av[8], bv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { const uint32_t o = (uint32_t)(k + 1) <= rem ? (uint32_t)(k + 1) : rem; av[k] = As[a1 - o]; bv[k] = Bs[b1 - o]; }
      const uint32_t n = rem < 8 ? rem : 8;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if ((uint32_t)k < n && !stop) {
          score += (int)sub2[(av[k] << 5) | bv[k]];
          if (score > best) { best = score; bestb1 = b1 - k - 1; }
          else if (best - score > X) stop = true;
        }
      a1 -= n; b1 -= n; rem -= n;
    }
  }
}

// One seed of UngappedBlast (ungappedblast.cpp:62-180): seed score, x-drop extension right then
// left, acceptance test.  nt: byte-SWAR - a run of matching letters is consumed per step (a match
// always raises the score, so inside a run the best is the run's end and the x-drop test cannot
// fire); aa: 8 letter pairs per LDS round trip, consumed strictly in order.
template <bool NT>
__device__ __forceinline__ bool extend_seed(const WaveCtx &c, const UgsDbView &db, int m2, int mm2, uint32_t apos,
                                            uint32_t bpos, uint32_t MinLength, uint32_t &oAlo, uint32_t &oBlo,
                                            uint32_t &oLen, int &oBest)
{
  const int w = db.hsp_w;
  const uint32_t LA = c.LA, LB = c.LB;
  const int X = db.xdrop2;
  int score = 0;
  if (NT) {
    // the two words are equal, so no pair of the seed is a mismatch: every pair scores a match unless one of its letters is not
    // A/C/G/T/U (both then carry the letter 0 in the word and the pair scores 0, setnucmx.cpp:11-99)
    const uint32_t ninv = (c.a_inv || c.b_inv) ? (uint32_t)__popc((nt_word(c.Ai, apos, w) | nt_word(c.Bi, bpos, w))) : 0u;
    score = ((int)w - (int)ninv) * m2;
  } else
    for (int k = 0; k < w; ++k) score += sscore<NT>(c, m2, mm2, c.As[apos + k], c.Bs[bpos + k]);
  int best = score;
  uint32_t b2 = bpos + w - 1, a2 = apos + w - 1, bestb2 = b2;
  uint32_t a1 = apos, b1 = bpos, bestb1 = b1;
  if (NT) {
    if (c.a_inv || c.b_inv) extend_nt_packed<true>(c.A2, c.Ai, c.B2, c.Bi, m2, mm2, X, LA, LB, a1, b1, a2, b2, score, best, bestb1, bestb2);
    else if (c.xlut) extend_nt_lut(c.A2, c.B2, c.xlut, LA, LB, a1, b1, a2, b2, best, bestb1, bestb2);
    else extend_nt_packed<false>(c.A2, c.Ai, c.B2, c.Bi, m2, mm2, X, LA, LB, a1, b1, a2, b2, score, best, bestb1, bestb2);
  } else extend_aa_bytes(c.s_sub2, c.As, c.Bs, X, LA, LB, a1, b1, a2, b2, score, best, bestb1, bestb2);
  const uint32_t Blo = bestb1, Bhi = bestb2, Len = Bhi - Blo + 1;
  const uint32_t Alo = apos - (bpos - bestb1);
  oAlo = Alo; oBlo = Blo; oLen = Len; oBest = best;
  return Len >= MinLength && best >= db.minscore2 && is_global_hsp(Alo, Blo, LA, LB);
}

// ungappedblast.cpp:8-211.  The reference walks target positions BPos upward, tries the (<= 8) query
// positions of that word in order and, on the first accepted HSP, jumps to BPos = Bhi+1.  Here the
// seeds (bpos, apos) are first listed in exactly that order into an LDS list (cheap: table lookups
// only), then extended 64 at a time - one seed per lane, so lanes are evenly loaded whatever the
// seed density - and the serial rule is replayed: the first accepting seed of a round wins, every
// seed at a target position <= its Bhi is dropped.
template <bool NT>
__device__ __forceinline__ void ungapped_blast(WaveCtx &c, const UgsDbView &db, uint32_t MinLength, unsigned long long *counters)
{
  const int lane = c.lane, w = db.hsp_w;
  const int m2 = c.s_sub2[0], mm2 = c.s_sub2[2];      // nt: 2*score(A,A), 2*score(A,C)
  const uint32_t LB = c.LB;
  uint32_t nh = 0;
  // is_global_hsp (hspfinder.cpp:594-636) depends on the seed's diagonal d = apos - bpos only (ALo - BLo = d, AR - BR = LA - LB - d):
  // a seed outside [dlo, dhi] can never be accepted, and an extension that is not accepted leaves no trace in the reference's
  // loop (ungappedblast.cpp:62-180) - such seeds are not even listed
  int dlo, dhi;
  {
    const int LAi = (int)c.LA, LBi = (int)LB;
    if (LAi <= LBi) { const int mg = LAi / 4 + 1; dlo = LAi - LBi - mg; dhi = mg; }
    else { const int mg = LBi / 4 + 1; dlo = -mg; dhi = mg + LAi - LBi; }
  }
  if (LB >= 2u * w && c.nwA > 0) {
    const uint32_t nwB = LB - w + 1;
    uint32_t scan = 0;          // next target position whose seeds are not listed yet
    uint32_t count = 0, idx = 0;
    for (;;) {
      // ---- list seeds until the list is comfortably full or the target is exhausted
#if UGS_ALIGN_CLOCKS == 4
      const unsigned long long tb0 = clock64();
#endif
      if (idx >= count) { idx = 0; count = 0; }
      while (scan < nwB && count + 64 * UGS_MAXREPS <= c.seed_cap) {
        if (NT && c.use_tab && c.bsh == 0 && nwB - scan > 64u) {
          // four instructions' worth of target positions at once: the word and table look-ups of all four are in flight together
          // and two prefix sums over packed 16-bit counts place them (a lane's count is <= MaxReps, a chunk's total <= 512)
          uint32_t bp[4], lo4[4], cn[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            bp[e] = scan + (uint32_t)e * 64u + (uint32_t)lane;
            lo4[e] = 0; cn[e] = 0;
          }
          uint32_t wd[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) wd[e] = bp[e] < nwB ? nt_word(c.B2, bp[e], w) : 0u;
#pragma unroll
          for (int e = 0; e < 4; ++e) if (bp[e] < nwB) { const uint32_t en = c.wstart[wd[e]]; lo4[e] = en & 0xfffu; cn[e] = en >> 12; }
          uint32_t okm[4];                                   // the query positions of the word (<= MaxReps) whose diagonal can be accepted
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            okm[e] = 0;
            for (uint32_t r = 0; r < cn[e]; ++r) { const int d = (int)(c.qsort[lo4[e] + r] & 0xffffu) - (int)bp[e]; okm[e] |= (d >= dlo && d <= dhi ? 1u : 0u) << r; }
            cn[e] = (uint32_t)__popc(okm[e]);
          }
          const uint32_t c01 = cn[0] | (cn[1] << 16), c23 = cn[2] | (cn[3] << 16);
          const uint32_t i01 = wave_incl_sum_u32(c01), i23 = wave_incl_sum_u32(c23);
          const uint32_t t01 = (uint32_t)__builtin_amdgcn_readlane((int)i01, 63), t23 = (uint32_t)__builtin_amdgcn_readlane((int)i23, 63);
          const uint32_t tot[4] = {t01 & 0xffffu, t01 >> 16, t23 & 0xffffu, t23 >> 16};
          const uint32_t total4 = tot[0] + tot[1] + tot[2] + tot[3];
          if (count + total4 <= c.seed_cap) {
            const uint32_t inc[4] = {i01 & 0xffffu, i01 >> 16, i23 & 0xffffu, i23 >> 16};
            uint32_t base = count;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              uint32_t off = base + inc[e] - cn[e];
              for (uint32_t m = okm[e]; m; m &= m - 1) { const uint32_t r = (uint32_t)__ffs((int)m) - 1u; c.seeds[off++] = (bp[e] << 16) | (c.qsort[lo4[e] + r] & 0xffffu); }
              base += tot[e];
            }
            count += total4;
            scan += 256;
            continue;
          }
        }
        const uint32_t bpos = scan + lane;
        uint32_t lo = 0, cnt = 0, wordv = 0;
        if (bpos < nwB) {
          uint32_t word = 0;
          if (NT) word = nt_word(c.B2, bpos, w);
          else for (int k = 0; k < w; ++k) word = word * db.alpha + c.s_hl[c.B[bpos + k] & 31];
          wordv = word;
          if (c.use_tab) { const uint32_t e = c.wstart[word >> c.bsh]; lo = e & 0xfffu; cnt = e >> 12; }
          else {
            const uint32_t want = word << 16;
            uint32_t hi = c.nwA;
            while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (c.qsort[mid] < want) lo = mid + 1; else hi = mid; }
            while (cnt < UGS_MAXREPS && lo + cnt < c.nwA && (c.qsort[lo + cnt] >> 16) == word) ++cnt;
          }
        }
        uint32_t okm1 = 0;
        if (c.use_tab && c.bsh) {
          // a bucket's entries in query-position order: the first MaxReps of THIS word are the word's positions (HSPFinder::SetA)
          uint32_t nm = 0;
          for (uint32_t r = 0; r < cnt; ++r) {
            const uint32_t e = c.qsort[lo + r];
            if ((e >> 16) == wordv) {
              const int d = (int)(e & 0xffffu) - (int)bpos;
              if (nm < UGS_MAXREPS) okm1 |= (d >= dlo && d <= dhi ? 1u : 0u) << r;
              ++nm;
            }
          }
        } else
        for (uint32_t r = 0; r < cnt; ++r) { const int d = (int)(c.qsort[lo + r] & 0xffffu) - (int)bpos; okm1 |= (d >= dlo && d <= dhi ? 1u : 0u) << r; }
        cnt = (uint32_t)__popc(okm1);
        const uint32_t incl = wave_incl_sum_u32(cnt);
        const uint32_t total = __builtin_amdgcn_readlane((int)incl, 63);
        uint32_t off = count + incl - cnt;
        for (uint32_t m = okm1; m; m &= m - 1) { const uint32_t r = (uint32_t)__ffs((int)m) - 1u; c.seeds[off++] = (bpos << 16) | (c.qsort[lo + r] & 0xffffu); }
        count += total;
        scan += 64;
      }
      lds_sync();
      if (idx >= count) { if (scan >= nwB) break; else continue; }
#if UGS_ALIGN_CLOCKS == 4
      const unsigned long long tb1 = clock64();
      if (lane == 0) { atomicAdd(&counters[UGS_CTR_T4], tb1 - tb0); atomicAdd(&counters[UGS_CTR_T6], 1ull); atomicAdd(&counters[UGS_CTR_T7], (unsigned long long)((count - idx) < 64u ? (count - idx) : 64u)); }
#endif
      // ---- extend one round of (up to) 64 seeds
      const uint32_t me = idx + lane;
      bool ok = false;
      uint32_t rAlo = 0, rBlo = 0, rLen = 0; int rBest = 0;
      uint32_t sd = 0;
      if (me < count) sd = c.seeds[me];
      bool first_done = false, first_ok = false;
      uint32_t uAlo = 0, uBlo = 0, uLen = 0; int uBest = 0;
      {
        // a round whose first seed shares its diagonal with many others (a homologous pair): that seed is extended by the whole
        // wave; accepted, it is the round's winner whatever the others give (first in list order); rejected, the round goes on without it
        const uint32_t sd0 = (uint32_t)rl((int)sd, 0);
        const int d0 = (int)(sd0 & 0xffffu) - (int)(sd0 >> 16);
        const bool same = me < count && (int)(sd & 0xffffu) - (int)(sd >> 16) == d0;
        if (__popcll(__ballot(same)) >= 8) {
          first_ok = extend_seed_coop<NT>(c, db, m2, mm2, sd0 & 0xffffu, sd0 >> 16, MinLength, uAlo, uBlo, uLen, uBest);
          first_done = true;
        }
      }
      int f = 0;
      uint32_t Alo, Blo, Len; int Best;
      if (first_ok) { Alo = uAlo; Blo = uBlo; Len = uLen; Best = uBest; }
      else {
        if (me < count && !(first_done && lane == 0))
          ok = extend_seed<NT>(c, db, m2, mm2, sd & 0xffffu, sd >> 16, MinLength, rAlo, rBlo, rLen, rBest);
        const uint64_t m = __ballot(ok);
#if UGS_ALIGN_CLOCKS == 4
        if (lane == 0) atomicAdd(&counters[UGS_CTR_T5], clock64() - tb1);
#endif
        if (!m) { idx += 64; continue; }
        f = __ffsll((long long)m) - 1;
        Alo = rl((int)rAlo, f); Blo = rl((int)rBlo, f); Len = rl((int)rLen, f);
        Best = rl(rBest, f);
      }
      if (nh < c.hsp_cap) {
        if (lane == 0) { c.hsps[nh].Loi = Alo; c.hsps[nh].Loj = Blo; c.hsps[nh].Len = Len; c.hsps[nh].Score2 = Best; }
        ++nh;
      } else if (lane == 0) atomicOr(&counters[UGS_CTR_ERR], (unsigned long long)UGS_ERR_HSPCAP);
      const uint32_t newB = Blo + Len;            // BPos = Bhi + 1
      if (newB >= scan) { scan = newB; idx = count; }     // everything listed so far is skipped
      else {
        // first listed seed with bpos >= newB (the list is ordered by bpos)
        uint32_t lo2 = idx + f + 1, hi2 = count;
        while (lo2 < hi2) { uint32_t mid = (lo2 + hi2) >> 1; if ((c.seeds[mid] >> 16) < newB) lo2 = mid + 1; else hi2 = mid; }
        idx = lo2;
      }
    }
  }
  if (lane == 0) c.ws->nhsp = nh;
  lds_sync();
}

// The quick test of a GROUP of candidates (nt, packed targets): does UngappedBlast find any HSP at all?  A query without a relative in the
// database walks max_rejects random candidates; each of them lists ~ 15 seeds inside the diagonal window and none extends to an HSP, so a
// pair on its own keeps a quarter of the wave busy for one extension round and pays the latency of its target fetch alone.  Whether a
// pair has an HSP does not depend on the order of its seeds (ungappedblast.cpp:62-180: a seed that is not accepted leaves no trace, the
// BPos = Bhi + 1 skip only ever follows an accepted one), so the seeds of up to UGS_GROUP targets are listed into one list (tagged with
// the member) and extended together, the targets' letters fetched together.  A member without an accepted seed is what the serial path
// calls a pair with no HSPs (globalalignmem.cpp:161-166, FailIfNoHSPs): a reject.  Any other member - an accepted seed, a letter that is
// not A/C/G/T/U, a length outside 2w .. 1024, a full seed list - is reported back ("maybe") and takes the full path, which decides.
// The members' 2-bit planes live in the target's class / score-code byte arrays (c.B, c.Bs), which hold nothing between two pairs.
#define UGS_GROUP 4
struct GroupArgs {           // (by value: the function is not inlined, the wave context stays in the caller's registers)
  const uint2 *pk; const uint32_t *A2; unsigned char *B, *Bs; const uint16_t *wstart; const uint32_t *qsort; uint32_t *seeds; const uint16_t *xlut;
  uint32_t seed_cap, LA, MinLength, gwt, nB; int w, X, m2, mm2, minscore2;
  unsigned long long *ctr;    // (-DUGS_ALIGN_CLOCKS=5, inlined builds only: the wave's accumulators - T4 fetch + planes, T5 seed listing, T6 extension rounds, T7 groups << 32 | seeds)
};
#ifndef UGS_GROUP_INLINE
#define UGS_GROUP_INLINE __forceinline__      // (as a function of its own, -DUGS_GROUP_INLINE=__noinline__, the call costs the candidate loop 2.5 ms per 1 M C2 queries)
#endif
__device__ UGS_GROUP_INLINE uint32_t group_filter(const GroupArgs c, uint32_t k, uint32_t n, uint64_t cto, uint32_t clen)
{
  const int lane = (int)(threadIdx.x & 63), w = c.w, X = c.X;
  const int m2 = c.m2, mm2 = c.mm2;
  const uint32_t LA = c.LA, MinLength = c.MinLength, gwt = c.gwt, nB = c.nB;
  uint32_t maybe = 0;
  uint32_t Lg[UGS_GROUP];
#if UGS_ALIGN_CLOCKS == 5
  const unsigned long long tg0 = clock64();
#endif
  uint2 lo[UGS_GROUP], hi[UGS_GROUP];
  uint32_t sh2[UGS_GROUP];
#pragma unroll
  for (int g = 0; g < UGS_GROUP; ++g) {
    Lg[g] = 0; lo[g] = make_uint2(0u, 0u); hi[g] = lo[g]; sh2[g] = 0;
    if ((uint32_t)g < n) {
      const uint64_t to = ((uint64_t)(uint32_t)rl((int)(cto >> 32), (int)(k + g)) << 32) | (uint32_t)rl((int)(uint32_t)cto, (int)(k + g));
      const uint32_t L = (uint32_t)rl((int)clen, (int)(k + g));
      if (L > 1024u || L < 2u * (uint32_t)w) maybe |= 1u << g;
      else {
        Lg[g] = L; sh2[g] = ((uint32_t)to & 15u) * 2u;
        const uint64_t w0 = (to >> 4) + (uint32_t)lane;
        if ((uint32_t)lane * 16u < L + 16u) { lo[g] = c.pk[w0]; hi[g] = c.pk[w0 + 1]; }
      }
    }
  }
  auto gbuf = [&](uint32_t g) -> uint32_t * { unsigned char *base = g < nB ? c.B + g * gwt : (c.Bs - 16) + (g - nB) * gwt; return (uint32_t *)base + 2; };
#pragma unroll
  for (int g = 0; g < UGS_GROUP; ++g) {
    if (Lg[g]) {
      uint32_t *B2 = gbuf((uint32_t)g);
      const uint32_t LB = Lg[g], nw = (LB + 15) >> 4, j = (uint32_t)lane;
      uint32_t w2 = __builtin_amdgcn_alignbit(hi[g].x, lo[g].x, sh2[g]), wi = __builtin_amdgcn_alignbit(hi[g].y, lo[g].y, sh2[g]);
      if (j + 1 == nw && (LB & 15u)) { const uint32_t m = (1u << (2u * (LB & 15u))) - 1u; w2 &= m; wi &= m; }
      if (j >= nw) { w2 = 0; wi = 0; }
      if (j < nw + 3) B2[j] = w2;
      if (nw + 3 > 64 && j < nw + 3 - 64) B2[64 + j] = 0;
      if (lane < 2) B2[-1 - lane] = 0;
      if (__ballot(wi != 0)) { maybe |= 1u << g; Lg[g] = 0; }
    }
  }
  lds_sync();
#if UGS_ALIGN_CLOCKS == 5
  const unsigned long long tg1 = clock64();
#endif
  // ---- the members' seeds, one list: member << 26 | bpos << 16 | apos
  uint32_t count = 0;
#pragma unroll
  for (int g = 0; g < UGS_GROUP; ++g) {
    if (Lg[g]) {
      const uint32_t *B2 = gbuf((uint32_t)g);
      const uint32_t LB = Lg[g], nwB = LB - w + 1;
      int dlo, dhi;
      {
        const int LAi = (int)LA, LBi = (int)LB;
        if (LAi <= LBi) { const int mg = LAi / 4 + 1; dlo = LAi - LBi - mg; dhi = mg; }
        else { const int mg = LBi / 4 + 1; dlo = -mg; dhi = mg + LAi - LBi; }
      }
      for (uint32_t scan = 0; scan < nwB; scan += 256) {
        uint32_t bp[4], lo4[4], cn[4], wd[4], okm[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { bp[e] = scan + (uint32_t)e * 64u + (uint32_t)lane; lo4[e] = 0; cn[e] = 0; }
#pragma unroll
        for (int e = 0; e < 4; ++e) wd[e] = bp[e] < nwB ? nt_word(B2, bp[e], w) : 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (bp[e] < nwB) { const uint32_t en = c.wstart[wd[e]]; lo4[e] = en & 0xfffu; cn[e] = en >> 12; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          okm[e] = 0;
          for (uint32_t r = 0; r < cn[e]; ++r) { const int d = (int)(c.qsort[lo4[e] + r] & 0xffffu) - (int)bp[e]; okm[e] |= (d >= dlo && d <= dhi ? 1u : 0u) << r; }
          cn[e] = (uint32_t)__popc(okm[e]);
        }
        const uint32_t c01 = cn[0] | (cn[1] << 16), c23 = cn[2] | (cn[3] << 16);
        const uint32_t i01 = wave_incl_sum_u32(c01), i23 = wave_incl_sum_u32(c23);
        const uint32_t t01 = (uint32_t)__builtin_amdgcn_readlane((int)i01, 63), t23 = (uint32_t)__builtin_amdgcn_readlane((int)i23, 63);
        const uint32_t tot[4] = {t01 & 0xffffu, t01 >> 16, t23 & 0xffffu, t23 >> 16};
        const uint32_t total4 = tot[0] + tot[1] + tot[2] + tot[3];
        if (count + total4 > c.seed_cap) { maybe |= 1u << g; break; }          // (the member's seeds listed so far are extended in vain)
        const uint32_t inc[4] = {i01 & 0xffffu, i01 >> 16, i23 & 0xffffu, i23 >> 16};
        uint32_t base = count;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          uint32_t off = base + inc[e] - cn[e];
          for (uint32_t m = okm[e]; m; m &= m - 1) { const uint32_t r = (uint32_t)__ffs((int)m) - 1u; c.seeds[off++] = ((uint32_t)g << 26) | (bp[e] << 16) | (c.qsort[lo4[e] + r] & 0xffffu); }
          base += tot[e];
        }
        count += total4;
      }
    }
  }
  lds_sync();
#if UGS_ALIGN_CLOCKS == 5
  const unsigned long long tg2 = clock64();
#endif
  // ---- extend them 64 at a time; the seeds of a member that has an accepted seed already are passed over
  const uint32_t all = (1u << n) - 1u;
  for (uint32_t idx = 0; idx < count && (maybe & all) != all; idx += 64) {
    const uint32_t me = idx + (uint32_t)lane;
    bool ok = false;
    uint32_t g = 0;
    if (me < count) {
      const uint32_t sd = c.seeds[me];
      g = sd >> 26;
      if (!((maybe >> g) & 1u)) {
        const uint32_t bpos = (sd >> 16) & 1023u, apos = sd & 0xffffu;
        const uint32_t *B2 = gbuf(g);
        const uint32_t LB = g == 0 ? Lg[0] : g == 1 ? Lg[1] : g == 2 ? Lg[2] : Lg[3];
        int score = w * m2, best = score;
        uint32_t b2 = bpos + w - 1, a2 = apos + w - 1, bestb2 = b2;
        uint32_t a1 = apos, b1 = bpos, bestb1 = b1;
        if (c.xlut) extend_nt_lut(c.A2, B2, c.xlut, LA, LB, a1, b1, a2, b2, best, bestb1, bestb2);
        else extend_nt_packed<false>(c.A2, c.A2, B2, B2, m2, mm2, X, LA, LB, a1, b1, a2, b2, score, best, bestb1, bestb2);
        const uint32_t Len = bestb2 - bestb1 + 1, Alo = apos - (bpos - bestb1);
        ok = Len >= MinLength && best >= c.minscore2 && is_global_hsp(Alo, bestb1, LA, LB);
      }
    }
    if (__ballot(ok)) {
#pragma unroll
      for (int gi = 0; gi < UGS_GROUP; ++gi) if (__ballot(ok && g == (uint32_t)gi)) maybe |= 1u << gi;
    }
  }
  lds_sync();
#if UGS_ALIGN_CLOCKS == 5
  { const unsigned long long tg3 = clock64(); c.ctr[0] += tg1 - tg0; c.ctr[1] += tg2 - tg1; c.ctr[2] += tg3 - tg2; c.ctr[3] += (1ull << 32) | count; }
#endif
  return maybe & all;
}

// The same filter for amino-acid databases (and any target kept as bytes): a member's letters are fetched from the byte array and kept as
// score codes (member 0 in the score-code array of the serial path, the others in the tail of the union region, where the filter's seed
// list then ends); words through the HSP letter of a score code, the bucketed word table as in ungapped_blast, extension by extend_aa_bytes.
struct GroupArgsAA {
  const uint8_t *seqs; const uint8_t *As; uint8_t *Bs0; uint8_t *tail; const uint16_t *wstart; const uint32_t *qsort; uint32_t *seeds;
  const uint8_t *s_cls, *s_sc, *s_hl; const int8_t *sub2;
  uint32_t seed_cap, LA, MinLength, gstride, bsh; int w, alpha, X, minscore2;
};
__device__ UGS_GROUP_INLINE uint32_t group_filter_aa(const GroupArgsAA c, uint32_t k, uint32_t n, uint64_t cto, uint32_t clen)
{
  const int lane = (int)(threadIdx.x & 63), w = c.w, X = c.X;
  const uint32_t LA = c.LA, MinLength = c.MinLength;
  uint32_t maybe = 0;
  uint32_t Lg[UGS_GROUP];
  uint32_t v[UGS_GROUP][4];
  auto gbuf = [&](uint32_t g) -> uint8_t * { return g == 0 ? c.Bs0 : c.tail + (g - 1u) * c.gstride; };
#pragma unroll
  for (int g = 0; g < UGS_GROUP; ++g) {
    Lg[g] = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[g][e] = 0;
    if ((uint32_t)g < n) {
      const uint64_t to = ((uint64_t)(uint32_t)rl((int)(cto >> 32), (int)(k + g)) << 32) | (uint32_t)rl((int)(uint32_t)cto, (int)(k + g));
      const uint32_t L = (uint32_t)rl((int)clen, (int)(k + g));
      if (L > 1024u || L < 2u * (uint32_t)w) maybe |= 1u << g;
      else {
        Lg[g] = L;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t o = (uint32_t)(e * 64 + lane) * 4;
          if (o < L) __builtin_memcpy(&v[g][e], c.seqs + to + o, 4);       // unaligned dword load (the DB buffer is padded)
        }
      }
    }
  }
#pragma unroll
  for (int g = 0; g < UGS_GROUP; ++g) {
    if (Lg[g]) {
      uint8_t *Bs = gbuf((uint32_t)g);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t o = (uint32_t)(e * 64 + lane) * 4;
        if (o < Lg[g]) {
          uint32_t sc4 = 0;
#pragma unroll
          for (uint32_t b = 0; b < 4; ++b) sc4 |= (uint32_t)c.s_sc[c.s_cls[(v[g][e] >> (8 * b)) & 0xffu] & 31] << (8 * b);
          *(uint32_t *)(Bs + o) = sc4;                                   // (codes behind the last letter are never read)
        }
      }
    }
  }
  lds_sync();
  // ---- the members' seeds, one list: member << 26 | bpos << 16 | apos
  uint32_t count = 0;
#pragma unroll
  for (int g = 0; g < UGS_GROUP; ++g) {
    if (Lg[g]) {
      const uint8_t *Bs = gbuf((uint32_t)g);
      const uint32_t LB = Lg[g], nwB = LB - w + 1;
      int dlo, dhi;
      {
        const int LAi = (int)LA, LBi = (int)LB;
        if (LAi <= LBi) { const int mg = LAi / 4 + 1; dlo = LAi - LBi - mg; dhi = mg; }
        else { const int mg = LBi / 4 + 1; dlo = -mg; dhi = mg + LAi - LBi; }
      }
      for (uint32_t scan = 0; scan < nwB; scan += 64) {
        const uint32_t bpos = scan + (uint32_t)lane;
        uint32_t lo = 0, cnt = 0, word = 0;
        if (bpos < nwB) {
          for (int q = 0; q < w; ++q) word = word * c.alpha + c.s_hl[Bs[bpos + q]];
          const uint32_t e = c.wstart[word >> c.bsh]; lo = e & 0xfffu; cnt = e >> 12;
        }
        uint32_t okm = 0;
        if (c.bsh) {
          // a bucket's entries in query-position order: the first MaxReps of THIS word are the word's positions (HSPFinder::SetA)
          uint32_t nm = 0;
          for (uint32_t r = 0; r < cnt; ++r) {
            const uint32_t e = c.qsort[lo + r];
            if ((e >> 16) == word) {
              const int d = (int)(e & 0xffffu) - (int)bpos;
              if (nm < UGS_MAXREPS) okm |= (d >= dlo && d <= dhi ? 1u : 0u) << r;
              ++nm;
            }
          }
        } else
          for (uint32_t r = 0; r < cnt; ++r) { const int d = (int)(c.qsort[lo + r] & 0xffffu) - (int)bpos; okm |= (d >= dlo && d <= dhi ? 1u : 0u) << r; }
        const uint32_t cn = (uint32_t)__popc(okm);
        const uint32_t incl = wave_incl_sum_u32(cn);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (count + total > c.seed_cap) { maybe |= 1u << g; break; }
        uint32_t off = count + incl - cn;
        for (uint32_t m = okm; m; m &= m - 1) { const uint32_t r = (uint32_t)__ffs((int)m) - 1u; c.seeds[off++] = ((uint32_t)g << 26) | (bpos << 16) | (c.qsort[lo + r] & 0xffffu); }
        count += total;
      }
    }
  }
  lds_sync();
  // ---- extend them 64 at a time; the seeds of a member that has an accepted seed already are passed over
  const uint32_t all = (1u << n) - 1u;
  for (uint32_t idx = 0; idx < count && (maybe & all) != all; idx += 64) {
    const uint32_t me = idx + (uint32_t)lane;
    bool ok = false;
    uint32_t g = 0;
    if (me < count) {
      const uint32_t sd = c.seeds[me];
      g = sd >> 26;
      if (!((maybe >> g) & 1u)) {
        const uint32_t bpos = (sd >> 16) & 1023u, apos = sd & 0xffffu;
        const uint8_t *Bs = gbuf(g);
        const uint32_t LB = g == 0 ? Lg[0] : g == 1 ? Lg[1] : g == 2 ? Lg[2] : Lg[3];
        int score = 0;
        for (int q = 0; q < w; ++q) score += (int)c.sub2[((uint32_t)c.As[apos + q] << 5) | Bs[bpos + q]];
        int best = score;
        uint32_t b2 = bpos + w - 1, a2 = apos + w - 1, bestb2 = b2;
        uint32_t a1 = apos, b1 = bpos, bestb1 = b1;
        extend_aa_bytes(c.sub2, c.As, Bs, X, LA, LB, a1, b1, a2, b2, score, best, bestb1, bestb2);
        const uint32_t Len = bestb2 - bestb1 + 1, Alo = apos - (bpos - bestb1);
        ok = Len >= MinLength && best >= c.minscore2 && is_global_hsp(Alo, bestb1, LA, LB);
      }
    }
    if (__ballot(ok)) {
#pragma unroll
      for (int gi = 0; gi < UGS_GROUP; ++gi) if (__ballot(ok && g == (uint32_t)gi)) maybe |= 1u << gi;
    }
  }
  lds_sync();
  return maybe & all;
}

// chainer.cpp:352-500 on lane 0 (HSP counts are tiny); csc layout: [bp_pos 2n][bp_idxlo 2n][prev n][cscore n][list n]
__device__ __forceinline__ void chain_lane0(WaveCtx &c)
{
  const uint32_t n = c.ws->nhsp;
  uint32_t nchain = 0;
  if (n) {
    uint32_t *bp_pos = c.csc, *bp_il = c.csc + 2 * c.hsp_cap, *prev = c.csc + 4 * c.hsp_cap;
    int32_t *cs = (int32_t *)(c.csc + 5 * c.hsp_cap);
    uint32_t *list = c.csc + 6 * c.hsp_cap;
    for (uint32_t i = 0; i < n; ++i) {
      bp_pos[2 * i] = c.hsps[i].Loi; bp_il[2 * i] = (i << 1) | 1u;
      bp_pos[2 * i + 1] = c.hsps[i].Loi + c.hsps[i].Len - 1; bp_il[2 * i + 1] = (i << 1);
    }
    // stable sort: Pos ascending, Lo before Hi, ties keep input order (glibc qsort = merge sort)
    for (uint32_t i = 1; i < 2 * n; ++i) {
      const uint32_t p = bp_pos[i], il = bp_il[i];
      uint32_t j = i;
      while (j > 0) {
        const uint32_t pp = bp_pos[j - 1], pil = bp_il[j - 1];
        const bool less = (p != pp) ? (p < pp) : (((il & 1) != (pil & 1)) ? ((il & 1) && !(pil & 1)) : false);
        if (!less) break;
        bp_pos[j] = pp; bp_il[j] = pil; --j;
      }
      bp_pos[j] = p; bp_il[j] = il;
    }
    for (uint32_t i = 0; i < n; ++i) prev[i] = 0xffffffffu;
    uint32_t nlist = 0;
    for (uint32_t b = 0; b < 2 * n; ++b) {
      if (!(bp_il[b] & 1)) continue;
      const uint32_t hi = bp_il[b] >> 1;
      const HSPd h = c.hsps[hi];
      int best = 0; uint32_t bestc = 0xffffffffu;
      for (uint32_t k = 0; k < nlist; ++k) {
        const uint32_t ci = list[k];
        const HSPd ch = c.hsps[ci];
        if (ch.Loi + ch.Len - 1 < h.Loi && ch.Loj + ch.Len - 1 < h.Loj && (bestc == 0xffffffffu || cs[ci] > best)) { bestc = ci; best = cs[ci]; }
      }
      list[nlist++] = hi;
      prev[hi] = bestc;
      cs[hi] = bestc == 0xffffffffu ? h.Score2 : cs[bestc] + h.Score2;
    }
    uint32_t opt = 0; int optscore = cs[0];
    for (uint32_t i = 1; i < n; ++i) if (cs[i] > optscore) { opt = i; optscore = cs[i]; }
    uint32_t len = 0;
    for (uint32_t i = opt; i != 0xffffffffu; i = prev[i]) ++len;
    uint32_t k = 1;
    for (uint32_t i = opt; i != 0xffffffffu; i = prev[i]) c.chain[len - k++] = i;
    nchain = len;
    // hspfinder.cpp:537-553 + hsp.h:102-126 IsStaggered
    const int LA = (int)c.LA, LB = (int)c.LB;
    for (uint32_t q = 0; q < nchain; ++q) {
      const HSPd h = uni(c.hsps[uni(c.chain[q])]);
      const int Hii = (int)(h.Loi + h.Len - 1), Hij = (int)(h.Loj + h.Len - 1);
      int gLA = (int)h.Loi - (int)h.Loj, gLB = (int)h.Loj - (int)h.Loi;
      int gRA = LA - Hii - 1 - (LB - Hij - 1), gRB = LB - Hij - 1 - (LA - Hii - 1);
      if (gLA < 0) gLA = 0;
      if (gLB < 0) gLB = 0;
      if (gRB < 0) gRB = 0;                     // TermGapRightA is not clamped in the reference
      const int GapA = gLA + gRA, GapB = gLB + gRB;
      if (GapA == 0 || GapB == 0) continue;
      const double r = (LA < LB ? (double)GapA / LA : (double)GapB / LB);
      if (r > 0.5) { nchain = 0; break; }
    }
  }
  c.ws->nchain = nchain;
}

// ---- run-length path assembly (lane 0 only); PathInfo::AppendPath/AppendMs (pathinfo.h:7-87)
__device__ __forceinline__ void put_run(WaveCtx &c, uint32_t i, uint32_t run)
{
  if (i < LRUNS) c.lds_runs[i] = run;
  else if (i < c.runs_cap) c.runs[i] = run;
  else c.ws->overflow = 1;
}
__device__ __forceinline__ uint32_t get_run(const WaveCtx &c, uint32_t i) { return i < LRUNS ? c.lds_runs[i] : c.runs[i]; }
__device__ __forceinline__ void put_rt(WaveCtx &c, uint32_t i, uint32_t run)
{
  if (i < LRUNS) c.lds_rt[i] = run;
  else if (i < c.runs_cap) c.runs[c.runs_cap + i] = run;
  else c.ws->overflow = 1;
}
__device__ __forceinline__ uint32_t get_rt(const WaveCtx &c, uint32_t i) { return i < LRUNS ? c.lds_rt[i] : c.runs[c.runs_cap + i]; }
__device__ __forceinline__ void push_run(WaveCtx &c, uint32_t op, uint32_t len)
{
  if (!len) return;
  WaveState *ws = c.ws;
  if (ws->cur_len && ws->cur_op == op) { ws->cur_len += len; return; }
  if (ws->cur_len) {
    put_run(c, ws->nruns, (ws->cur_len << 2) | ws->cur_op);
    ++ws->nruns;
  }
  ws->cur_op = op; ws->cur_len = len;
}
__device__ __forceinline__ void flush_runs(WaveCtx &c)
{
  WaveState *ws = c.ws;
  if (ws->cur_len) {
    put_run(c, ws->nruns, (ws->cur_len << 2) | ws->cur_op);
    ++ws->nruns; ws->cur_len = 0;
  }
}

// wave-wide inclusive prefix max and shift by one lane with DPP (row shifts + row broadcasts: no LDS crossbar round trips; r04: the row
// sweep of viterbi_hole made eight ds_bpermute round trips per row)
__device__ __forceinline__ int wave_incl_max_i32(int v)
{
  int x;
  x = __builtin_amdgcn_update_dpp(NEG, v, 0x111, 0xf, 0xf, false); v = x > v ? x : v;      // row_shr:1
  x = __builtin_amdgcn_update_dpp(NEG, v, 0x112, 0xf, 0xf, false); v = x > v ? x : v;      // row_shr:2
  x = __builtin_amdgcn_update_dpp(NEG, v, 0x114, 0xf, 0xf, false); v = x > v ? x : v;      // row_shr:4
  x = __builtin_amdgcn_update_dpp(NEG, v, 0x118, 0xf, 0xf, false); v = x > v ? x : v;      // row_shr:8
  x = __builtin_amdgcn_update_dpp(NEG, v, 0x142, 0xa, 0xf, false); v = x > v ? x : v;      // row_bcast:15 -> rows 1, 3
  x = __builtin_amdgcn_update_dpp(NEG, v, 0x143, 0xc, 0xf, false); v = x > v ? x : v;      // row_bcast:31 -> rows 2, 3
  return v;
}
// lane l gets v of lane l - 1, lane 0 gets `first`
__device__ __forceinline__ int wave_shr1(int v, int first) { return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false); }

// diagbox.h:150-171
__device__ __forceinline__ void get_range_j(uint32_t LA, uint32_t LB, uint32_t dlo, uint32_t dhi, uint32_t i,
                                            uint32_t &Startj, uint32_t &Endj)
{
  uint32_t s = (dlo + i >= LA) ? dlo + i - LA : 0;
  if (s >= LB) s = LB - 1;
  uint32_t e = (dhi + i + 1 >= LA) ? dhi + i + 1 - LA : 0;
  if (e > LB) e = LB;
  Startj = s; Endj = e;
}

// ViterbiFastMainDiagMem + ViterbiFastBandMem + TraceBackBitMem on the hole A[a0..a0+LA) x B[b0..b0+LB);
// appends the path to the run list.  All lanes participate; lane 0 does the traceback.
__device__ __forceinline__ void viterbi_hole(WaveCtx &c, uint32_t a0, uint32_t LA, uint32_t b0, uint32_t LB, uint32_t band,
                             const Pen &P, unsigned long long *counters)
{
  const int lane = c.lane;
#if UGS_ALIGN_CLOCKS == 3
  const unsigned long long tv0 = clock64();
#endif
  uint32_t dlo = LA < LB ? LA : LB, dhi = LA > LB ? LA : LB;
  if (dlo > band) dlo -= band; else dlo = 1;
  dhi += band;
  if (dhi > LA + LB - 1) dhi = LA + LB - 1;
  const uint32_t stride = (dhi - dlo + 1) + 3;
  // Where the traceback bytes live: the 1 KB LDS area of the wave for small holes; for larger ones the part of the union region (seed
  // list / DP rows / chainer scratch - idle here except for the rows) that the two DP rows of THIS hole leave free: they need LB + 8
  // entries each, not max_tlen + 8.  r04: the average hole of a C2 hit is 35 x 35 = 1.3 KB of traceback bytes and went through HBM scratch
  // (a store round trip per row behind the fence, a dependent global load per traceback step: 178 k cycles per hole); only holes beyond
  // ~50 x 50 still do.
  int32_t *Mrow = c.Mrow, *Drow = c.Drow;
  uint8_t *TB;
  const uint64_t tb_bytes = (uint64_t)(LA + 1) * stride;
  if (tb_bytes <= LTB) TB = c.lds_tb;
  else {
    const uint32_t row_words = LB + 8u;                               // Mrow[-1 .. LB], Drow[0 .. LB]
    const uint64_t tb_off = (((uint64_t)4 + 2ull * row_words) * 4 + 15) & ~15ull;      // bytes from the union base (Mrow starts 4 words in)
    if (tb_off + tb_bytes <= (uint64_t)c.union_words * 4) {
      Drow = Mrow - 4 + 4 + row_words;                                // rows packed at the front of the region
      TB = (uint8_t *)(Mrow - 4) + tb_off;
    } else TB = c.tb;
  }
  // the traceback bytes are addressed as LDS where they are LDS (a generic pointer makes every store of the row sweep and every load of
  // the single-lane traceback a flat instruction, whose latency the fences of the sweep and the traceback's dependent chain then expose)
  const bool tb_lds = TB != c.tb;
  typedef uint8_t __attribute__((address_space(3))) *lds8;
  const lds8 TBl = (lds8)(uintptr_t)(uint32_t)(uintptr_t)TB;
  auto tb_st = [&](uint64_t off, uint8_t v) { if (tb_lds) TBl[(uint32_t)off] = v; else c.tb[off] = v; };
  auto tb_ld = [&](uint64_t off) -> uint8_t { return tb_lds ? TBl[(uint32_t)off] : c.tb[off]; };
  unsigned long long cells = 0;
  int dEnd = NEG;                                                  // Drow[LB]: touched by the end-of-row special case only - a wave-uniform register
  uint32_t Startj = 0, Endj = 0;                                   // (after the sweep: the range of the last row)
  int carryI = NEG, FinalM = NEG;
  if (LB <= 64u) {
    // ---- a hole at most 64 columns wide (the usual one): lane = COLUMN for the whole hole.  The two DP rows are registers (no LDS round
    // trip and no fence per row), the column's target letter is read once, the score look-up of a row is issued before the row's
    // recurrences (it depends on the letters only) and the next row's query letter travels a row ahead.  Same recurrences, same traceback
    // bytes in the same layout as the general sweep below (viterbifastbandmem.cpp:12-204).
    const uint32_t jc = (uint32_t)lane;
    int M = NEG, D = NEG;                                          // Mrow[jc], Drow[jc]
    const uint8_t bl = jc < LB ? c.B[b0 + jc] : (uint8_t)0;
    uint8_t a_next = c.A[a0];
    for (uint32_t i = 0; i < LA; ++i) {
      const uint8_t a = a_next;
      if (i + 1 < LA) a_next = c.A[a0 + i + 1];
      get_range_j(LA, LB, dlo, dhi, i, Startj, Endj);
      if (Endj == 0) continue;
      const bool act = jc >= Startj && jc < Endj;
      const int sc = act ? score2(c, a, bl) : 0;
      const int OpenA = i == 0 ? P.LOpenA : P.OpenA, ExtA = i == 0 ? P.LExtA : P.ExtA;
      const uint64_t rowo = (uint64_t)i * stride;
      if (Startj > 0 && lane == 0) tb_st(rowo, TB_IM);           // (the entry of column Startj - 1)
      cells += Endj - Startj;
      const int oldM = act ? M : NEG;
      const int oldD = act ? D : NEG;
      int saved = wave_shr1(M, NEG);                             // Mrow[jc - 1] as the previous row left it
      if (i == 0 && jc == Startj) saved = 0;
      const int mi = act ? sat_add(saved, OpenA) : NEG;
      int v = mi <= NEGT ? NEG : mi - lane * ExtA;
      v = wave_incl_max_i32(v);
      const int Iout = v <= NEGT ? NEG : v + lane * ExtA;
      const int Iprev = wave_shr1(Iout, NEG);
      uint8_t bits = 0;
      int xM = saved;
      if (oldD > xM) { xM = oldD; bits = TB_DM; }
      if (Iprev > xM) { xM = Iprev; bits = TB_IM; }
      const int newM = sat_add(xM, sc);
      const int ob = (jc == 0) ? P.LOpenB : P.OpenB, eb = (jc == 0) ? P.LExtB : P.ExtB;
      const int md = sat_add(saved, ob);
      int nd = sat_add(oldD, eb);
      if (md >= nd) { nd = md; bits |= TB_MD; }
      const int iext = sat_add(Iprev, ExtA);
      if (mi >= iext) bits |= TB_MI;
      if (act) { M = newM; D = nd; tb_st(rowo + (jc - Startj + 1), bits); }
      const int carryM = rl(oldM, (int)Endj - 1);
      {                                                           // "Special case for end of Drow[]"
        uint8_t tbe = 0;
        const int mdl = sat_add(carryM, P.ROpenB);
        int dl = sat_add(dEnd, P.RExtB);
        if (mdl >= dl) { dl = mdl; tbe = TB_MD; }
        dEnd = dl;
        if (lane == 0) tb_st(rowo + stride - 1, tbe);
      }
    }
    // last row of DPI (viterbifastbandmem.cpp:186-204), strict '>'
    get_range_j(LA, LB, dlo, dhi, LA - 1, Startj, Endj);
    const uint64_t lasto = (uint64_t)LA * stride;
    cells += LB;
    {
      const bool act = jc >= Startj && jc < Endj;
      int Ms = wave_shr1(M, NEG);
      if (jc == Startj) Ms = NEG;                                 // (Mrow[Startj - 1] = NEG)
      const int mi = act ? sat_add(Ms, P.ROpenA) : NEG;
      int v = mi <= NEGT ? NEG : mi - lane * P.RExtA;
      v = wave_incl_max_i32(v);
      const int Iout = v <= NEGT ? NEG : v + lane * P.RExtA;
      const int Iprev = wave_shr1(Iout, NEG);
      const int iext = sat_add(Iprev, P.RExtA);
      if (act) tb_st(lasto + (jc - Startj + 1), (mi > iext) ? TB_MI : 0);
      if (Endj > Startj) carryI = rl(Iout, (int)Endj - 1);
    }
    FinalM = rl(M, (int)LB - 1);
    wave_sync();
  } else {
    for (uint32_t j = lane; j <= LB + 1; j += 64) { Mrow[(int)j - 1] = NEG; if (j <= LB) Drow[j] = NEG; }
    wave_sync();
    for (uint32_t i = 0; i < LA; ++i) {
      uint32_t Startj, Endj;
      get_range_j(LA, LB, dlo, dhi, i, Startj, Endj);
      if (Endj == 0) continue;
      const uint8_t a = c.A[a0 + i];
      const int OpenA = i == 0 ? P.LOpenA : P.OpenA, ExtA = i == 0 ? P.LExtA : P.ExtA;
      int carryM = (i == 0) ? 0 : (Startj == 0 ? NEG : Mrow[(int)Startj - 1]);
      int carryI = NEG;
      const uint64_t rowo = (uint64_t)i * stride;
      if (Startj > 0 && lane == 0) tb_st(rowo, TB_IM);             // (the entry of column Startj - 1)
      cells += Endj - Startj;
      for (uint32_t j0 = Startj; j0 < Endj; j0 += 64) {
        const uint32_t j = j0 + lane;
        const bool act = j < Endj;
        const int oldM = act ? Mrow[j] : NEG;
        const int oldD = act ? Drow[j] : NEG;
        const int saved = wave_shr1(oldM, carryM);
        const int mi = act ? sat_add(saved, OpenA) : NEG;
        // in-row insert recurrence I[k] = max(mi[k], I[k-1]+ExtA) as a max-plus prefix scan
        int v = mi <= NEGT ? NEG : mi - lane * ExtA;
        v = wave_incl_max_i32(v);
        const int fromscan = v <= NEGT ? NEG : v + lane * ExtA;
        const int fromcarry = carryI <= NEGT ? NEG : carryI + (lane + 1) * ExtA;
        const int Iout = fromscan > fromcarry ? fromscan : fromcarry;
        const int Iprev = wave_shr1(Iout, carryI);
        uint8_t bits = 0;
        int xM = saved;
        if (oldD > xM) { xM = oldD; bits = TB_DM; }
        if (Iprev > xM) { xM = Iprev; bits = TB_IM; }
        const int sc = act ? score2(c, a, c.B[b0 + j]) : 0;
        const int newM = sat_add(xM, sc);
        const int ob = (j == 0) ? P.LOpenB : P.OpenB, eb = (j == 0) ? P.LExtB : P.ExtB;
        const int md = sat_add(saved, ob);
        int nd = sat_add(oldD, eb);
        if (md >= nd) { nd = md; bits |= TB_MD; }
        const int iext = sat_add(Iprev, ExtA);
        if (mi >= iext) bits |= TB_MI;
        if (act) { Mrow[j] = newM; Drow[j] = nd; tb_st(rowo + (j - Startj + 1), bits); }
        const int lastl = (Endj - j0) >= 64 ? 63 : (int)(Endj - j0) - 1;
        carryM = rl(oldM, lastl);
        carryI = rl(Iout, lastl);
        wave_sync();
      }
      {                                                             // "Special case for end of Drow[]" (scalar: no LDS round trip, no second fence per row)
        uint8_t tbe = 0;
        const int md = sat_add(carryM, P.ROpenB);
        int dl = sat_add(dEnd, P.RExtB);
        if (md >= dl) { dl = md; tbe = TB_MD; }
        dEnd = dl;
        if (lane == 0) tb_st(rowo + stride - 1, tbe);
      }
    }
    // last row of DPI (viterbifastbandmem.cpp:186-204), strict '>'
    get_range_j(LA, LB, dlo, dhi, LA - 1, Startj, Endj);
    const uint64_t lasto = (uint64_t)LA * stride;
    if (lane == 0) Mrow[(int)Startj - 1] = NEG;
    wave_sync();
    cells += LB;
    for (uint32_t j0 = Startj; j0 < Endj; j0 += 64) {
      const uint32_t j = j0 + lane;
      const bool act = j < Endj;
      const int mi = act ? sat_add(Mrow[(int)j - 1], P.ROpenA) : NEG;
      int v = mi <= NEGT ? NEG : mi - lane * P.RExtA;
      v = wave_incl_max_i32(v);
      const int fromscan = v <= NEGT ? NEG : v + lane * P.RExtA;
      const int fromcarry = carryI <= NEGT ? NEG : carryI + (lane + 1) * P.RExtA;
      const int Iout = fromscan > fromcarry ? fromscan : fromcarry;
      const int Iprev = wave_shr1(Iout, carryI);
      const int iext = sat_add(Iprev, P.RExtA);
      if (act) tb_st(lasto + (j - Startj + 1), (mi > iext) ? TB_MI : 0);
      const int lastl = (Endj - j0) >= 64 ? 63 : (int)(Endj - j0) - 1;
      carryI = rl(Iout, lastl);
    }
    wave_sync();
    FinalM = rl(Mrow[LB - 1], 0);
  }
#if UGS_ALIGN_CLOCKS == 3
  const unsigned long long tv1 = clock64();
#endif
  {
    // The traceback is one dependent chain.  Every lane walks it (the same bytes: LDS broadcast reads, made wave-uniform with
    // readfirstlane), so the state and the index arithmetic live in SCALAR registers and issue on the scalar unit - as lane-0-only code it
    // was a chain of ~ 40 vector instructions per step (1 000 cycles a step, a third of an amino-acid pair's time); lane 0 alone writes the runs
    if (lane == 0) atomicAdd(&counters[UGS_CTR_CELLS], cells);
    const int FinalD = dEnd, FinalI = carryI;
    int Score = FinalM; uint32_t State = 0;                       // 0=M 1=D 2=I
    if (FinalD > Score) { Score = FinalD; State = 1; }
    if (FinalI > Score) { Score = FinalI; State = 2; }
    // traceback: reversed runs go to the upper half of the run buffer, then get pushed forward
    uint32_t nrt = 0, curop = 3, curlen = 0;
    uint32_t i = LA, j = LB;
    const uint32_t sLast = Startj;
    auto tbget = [&](uint32_t ti, uint32_t tj) -> uint8_t {
      if (tj == LB && ti < LA) return (uint8_t)__builtin_amdgcn_readfirstlane((int)tb_ld((uint64_t)ti * stride + stride - 1));
      uint32_t s, e;
      if (ti >= LA) s = sLast; else get_range_j(LA, LB, dlo, dhi, ti, s, e);
      const int idx = (int)tj - (int)s + 1;
      if (idx < 0 || idx >= (int)stride - 1) return 0;
      return (uint8_t)__builtin_amdgcn_readfirstlane((int)tb_ld((uint64_t)ti * stride + idx));
    };
    while (i != 0 || j != 0) {
#if UGS_TB_RUNS
      // A run of match columns in ONE trip (r5): in state M the chain goes down the diagonal for as long as the bytes on it say "M came
      // from M" - at 0.97 identity for dozens of cells - and lane k can look at the k-th cell of the diagonal by itself: one LDS read,
      // one ballot, and the walk moves to the first cell that says otherwise (or 64 cells on, or to the edge of the matrix).
      if (State == 0 && i != 0 && j != 0) {
        const uint32_t n0 = i < j ? i : j, n = n0 < 64u ? n0 : 64u;
        const uint32_t k = (uint32_t)lane;
        const bool in = k < n;
        const uint32_t ti = in ? i - 1u - k : 0u, tj = in ? j - 1u - k : 0u;          // (ti < LA, tj < LB: the common case of tbget)
        uint32_t sj, ej;
        get_range_j(LA, LB, dlo, dhi, ti, sj, ej);
        const int idx = (int)tj - (int)sj + 1;
        uint8_t tv = 0;
        if (in && idx >= 0 && idx < (int)stride - 1) tv = tb_ld((uint64_t)ti * stride + (uint32_t)idx);
        const uint64_t mleave = __ballot(in && (tv & (TB_DM | TB_IM)) != 0);
        uint32_t steps = n, nstate = 0;
        if (mleave) {
          const int k0 = __ffsll((long long)mleave) - 1;
          const uint32_t t0 = (uint32_t)rl((int)tv, k0);
          steps = (uint32_t)k0 + 1u; nstate = (t0 & TB_DM) ? 1u : 2u;
        }
        if (curop == 0) curlen += steps;
        else { if (curlen) { if (lane == 0) put_rt(c, nrt, (curlen << 2) | curop); ++nrt; } curop = 0; curlen = steps; }
        i -= steps; j -= steps; State = nstate;
        continue;
      }
#endif
      if (State == curop) ++curlen;
      else { if (curlen) { if (lane == 0) put_rt(c, nrt, (curlen << 2) | curop); ++nrt; } curop = State; curlen = 1; }
      uint8_t t;
      if (State == 0) { t = tbget(i - 1, j - 1); State = (t & TB_DM) ? 1 : ((t & TB_IM) ? 2 : 0); --i; --j; }
      else if (State == 1) { t = tbget(i - 1, j); State = (t & TB_MD) ? 0 : 1; --i; }
      else { t = tbget(i, j - 1); State = (t & TB_MI) ? 0 : 2; --j; }
    }
    if (curlen) { if (lane == 0) put_rt(c, nrt, (curlen << 2) | curop); ++nrt; }
    if (nrt > c.runs_cap) nrt = c.runs_cap;
    if (lane == 0) for (int k = (int)nrt - 1; k >= 0; --k) { const uint32_t r = get_rt(c, (uint32_t)k); push_run(c, r & 3, r >> 2); }
  }
  wave_sync();
#if UGS_ALIGN_CLOCKS == 3
  if (lane == 0) { atomicAdd(&counters[UGS_CTR_T4], 1ull); atomicAdd(&counters[UGS_CTR_T5], (unsigned long long)LA << 32 | LB); atomicAdd(&counters[UGS_CTR_T6], tv1 - tv0); atomicAdd(&counters[UGS_CTR_T7], clock64() - tv1); }
#endif
}

// globalalignmem.cpp:70-112 AlignHSPMem on a hole
__device__ __forceinline__ void align_hole(WaveCtx &c, const UgsDbView &db, uint32_t Loi, uint32_t Loj, uint32_t Leni, uint32_t Lenj,
                           unsigned long long *counters)
{
  if (Leni == 0) { if (c.lane == 0) push_run(c, 2, Lenj); wave_sync(); return; }
  if (Lenj == 0) { if (c.lane == 0) push_run(c, 1, Leni); wave_sync(); return; }
  const bool LeftA = Loi == 0, LeftB = Loj == 0, RightA = Loi + Leni == c.LA, RightB = Loj + Lenj == c.LB;
  Pen P;
  P.OpenA = P.OpenB = db.open2; P.ExtA = P.ExtB = db.ext2;
  P.LOpenA = LeftA ? db.topen2 : db.open2;  P.LExtA = LeftA ? db.text2 : db.ext2;
  P.LOpenB = LeftB ? db.topen2 : db.open2;  P.LExtB = LeftB ? db.text2 : db.ext2;
  P.ROpenA = RightA ? db.topen2 : db.open2; P.RExtA = RightA ? db.text2 : db.ext2;
  P.ROpenB = RightB ? db.topen2 : db.open2; P.RExtB = RightB ? db.text2 : db.ext2;
  viterbi_hole(c, Loi, Leni, Loj, Lenj, (uint32_t)db.band, P, counters);
}

// PAIR: the pair filters, -abskew, -fulldp and -gaforce are compiled into an instantiation of their own, so that the usual launch keeps the
// code (and register allocation) it was tuned with
// phase clocks (UGS_PHASE_CLOCKS report): reading the clock waits for every outstanding LDS / scalar-memory operation of the wave,
// four times per pair - compiled in only for tuning builds (-DUGS_ALIGN_CLOCKS=1)
// (-DUGS_ALIGN_CLOCKS=4: inside ungapped_blast - T4 seed listing, T5 extension rounds, T6 rounds, T7 seeds extended; the per-round atomics slow the kernel several times: ratios only)
#define ACLK() (UGS_ALIGN_CLOCKS == 1 ? clock64() : 0ull)
#define ACLK2() (UGS_ALIGN_CLOCKS == 2 ? clock64() : 0ull)      // finer clocks inside the post-HSP part: T4 chain, T5 classes + HSP identity, T6 holes, T7 FillLo + hit
// NT: nucleotide / amino-acid databases have instantiations of their own - the packed-letter paths, the group filter and the extension
// table exist only in the first, the bucketed word table and the BLOSUM look-ups only in the second, and neither carries the other's registers
template <bool PAIR, bool NT>
#ifndef UGS_ALIGN_WGS
#define UGS_ALIGN_WGS 4
#endif
__global__ __launch_bounds__(256, UGS_ALIGN_WGS) void k_align(UgsDbView db, UgsBatchView bv, uint32_t hsp_cap, uint32_t wave_lds, uint32_t seed_cap)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // (the wave index is wave-uniform, but derived from threadIdx the compiler holds it - and every per-wave LDS pointer and the two
  // global scratch pointers made from it - in VECTOR registers: 60 of the kernel's 95 spilled VGPRs were those; r5)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wpb = blockDim.x >> 6;
  // ---- workgroup-shared tables
  uint8_t *s_cls = smem;                       // 256
  uint8_t *s_hl = smem + 256;                  // 32
  int8_t *s_sub2 = (int8_t *)(smem + 288);     // 1024
  uint64_t *s_match = (uint64_t *)(smem + 1312);   // 64*8 = 512
  uint8_t *s_comp = smem + 1824;               // 256  -> 2080
  uint8_t *s_sc = smem + 2080;                 // 32   -> 2112: letter -> score code
  const UgsTables *tab = db.tab;
  for (int k = tid; k < 256; k += blockDim.x) { s_cls[k] = tab->cls[k]; s_comp[k] = tab->comp[k]; }
  for (int k = tid; k < 1024; k += blockDim.x) s_sub2[k] = tab->sub2[k];
  for (int k = tid; k < 64; k += blockDim.x) s_match[k] = tab->match[k];
  for (int k = tid; k < 32; k += blockDim.x) {
    s_hl[k] = (k < 26) ? tab->hsp_letter['A' + k] : 0;
    if (NT) s_sc[k] = (k < 26 && tab->udb_letter['A' + k] != 0xff) ? tab->hsp_letter['A' + k] : 4;
    else s_sc[k] = (k < 26) ? (uint8_t)k : 31;
  }
  // the extension table (extend_nt_lut): entry (d / 2, four mismatch bits) by simulating the serial x-drop rule
  uint16_t *s_xlut = (uint16_t *)(smem + 2112);
  bool xlut_ok = false;
  {
    const int m2 = tab->sub2[0], mm2 = tab->sub2[2], X = db.xdrop2;      // nt: 2 * score(A, A), 2 * score(A, C)
    xlut_ok = NT && m2 > 0 && mm2 < 0 && !(m2 & 1) && !(mm2 & 1) && !(X & 1) && X >= 0 && X <= 2 * (UGS_XLUT_D - 1) && m2 <= 30;
    if (xlut_ok)
      for (int k = tid; k < UGS_XLUT_D * 16; k += blockDim.x) {
        int cur = -2 * (k >> 4), gain = 0, o = 0; bool stop = false;      // cur = score - best
        for (int j = 0; j < 4 && !stop; ++j) {
          cur += ((k >> j) & 1) ? mm2 : m2;
          if (cur > 0) { gain += cur; cur = 0; o = j + 1; }
          else if (-cur > X) stop = true;
        }
        const int dn = stop ? 0 : (-cur) / 2;
        s_xlut[k] = (uint16_t)((stop ? 0x8000 : 0) | (o << 12) | ((gain / 2) << 6) | dn);
      }
  }
  __syncthreads();

  // ---- per-wave carve
  const uint32_t maxq = (bv.max_qlen + 15u) & ~15u, maxt = (db.max_tlen + 15u) & ~15u;
  uint32_t q2 = 64; while (q2 < maxq) q2 <<= 1;
  unsigned char *wb = smem + UGS_ALIGN_HDR + (size_t)wave * wave_lds;
  WaveCtx c;
  size_t off = 0;
  c.ws = (WaveState *)(wb + off); off += 32;
  c.A = wb + off; off += maxq;
  c.B = wb + off; off += maxt;
  c.As = wb + off + 16; off += maxq + 32;
  c.Bs = wb + off + 16; off += maxt + 32;
  {
    const size_t wa = (((size_t)maxq / 16 + 6) * 4 + 15) & ~(size_t)15, wt = (((size_t)maxt / 16 + 6) * 4 + 15) & ~(size_t)15;
    c.A2 = (uint32_t *)(wb + off) + 2; off += wa; c.Ai = (uint32_t *)(wb + off) + 2; off += wa;
    c.B2 = (uint32_t *)(wb + off) + 2; off += wt; c.Bi = (uint32_t *)(wb + off) + 2; off += wt;
  }
  c.nwords = (uint32_t)db.hsp_words;
  c.wstart = nullptr; c.bsh = 0; c.use_tab = false;
  while (((uint32_t)db.hsp_words - 1u) >> c.bsh >= 1024u) ++c.bsh;           // (aa: 8000 words -> 1000 buckets of 8)
  if (bv.max_qlen < 4096) { c.wstart = (uint16_t *)(wb + off); off += ((size_t)(((uint32_t)db.hsp_words - 1u) >> c.bsh) * 2 + 2 + 15) & ~(size_t)15; }
  c.lds_runs = (uint32_t *)(wb + off); off += LRUNS * 4;
  c.lds_rt = (uint32_t *)(wb + off); off += LRUNS * 4;
  c.lds_tb = wb + off; off += LTB;
  c.nt = NT;
  c.a_inv = false; c.b_inv = false;
  // union region: the seed list lives only inside UngappedBlast, the chainer scratch only inside the chainer, the DP
  // rows only in the holes after it
  const size_t uo = std::max(2 * ((size_t)maxt + 8) * 4, (size_t)hsp_cap * 7 * 4);
  const size_t us = std::max((size_t)seed_cap * 4, uo);
  c.union_words = (uint32_t)(us / 4);
  // the sorted query words need a power-of-two array only for the bitonic fallback; when every query of the batch
  // takes the counting sort (word table present and its scratch fits the union region) maxq entries do
  const bool always_counting = c.wstart != nullptr && c.bsh == 0 && (uint64_t)maxq * 3 / 2 + 16 <= c.union_words;
  c.qsort = (uint32_t *)(wb + off); off += (size_t)(always_counting ? maxq : q2) * 4;
  c.hsps = (HSPd *)(wb + off); off += (size_t)hsp_cap * sizeof(HSPd);
  c.chain = (uint32_t *)(wb + off); off += (size_t)hsp_cap * 4;
  {
    unsigned char *u = wb + off;
    c.seeds = (uint32_t *)u; c.seed_cap = seed_cap;
    c.Mrow = (int32_t *)u + 4;
    c.Drow = (int32_t *)(u + ((size_t)maxt + 8) * 4);
    c.csc = (uint32_t *)u;
    off += (us + 15) & ~(size_t)15;
  }
  c.s_cls = s_cls; c.s_sub2 = s_sub2; c.s_match = s_match; c.s_hl = s_hl; c.xlut = xlut_ok ? s_xlut : nullptr;
  c.lane = lane; c.hsp_cap = hsp_cap;
  const uint32_t gw = blockIdx.x * wpb + wave, nw = gridDim.x * wpb;
  c.tb = bv.tb + (uint64_t)gw * bv.tb_stride;
  c.runs = bv.runs + (uint64_t)gw * bv.runs_stride;
  c.runs_cap = bv.runs_stride / 2;

  const uint32_t units = bv.nq * bv.nstrand, K = bv.K;
  const uint32_t max_acc = (uint32_t)db.max_accepts, max_rej = (uint32_t)db.max_rejects;
  unsigned long long *ctr = bv.counters;

  unsigned long long ta0 = 0, ta1 = 0, ta2 = 0, ta3 = 0, tq;
  unsigned long long gclk[4] = {0, 0, 0, 0};
  unsigned long long w_tletters = 0, w_pairs = 0, w_grouped = 0;
  // units are handed out dynamically: a query without a hit walks all its candidates (32x the work of a query that
  // accepts its first one), so a static stride leaves most waves idle while the unlucky ones finish
  (void)gw; (void)nw;
  for (;;) {
    uint32_t unit = 0;
    if (lane == 0) unit = (uint32_t)atomicAdd(&ctr[UGS_CTR_NEXT_UNIT], 1ull);
    unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit);
    // -termid / -termidd (terminator.cpp:66-87) look at the hits of BOTH strands (one HitMgr per query): with them a wave takes a
    // whole query and walks its strands one after the other
    const uint32_t tflags = PAIR ? (db.align_flags & (UGS_A_TERMID | UGS_A_TERMIDD)) : 0u;
    // deep walks (ugs_deep.hip): a continuation pass takes the parked units from walk_units and walks on through their complete lists
    const bool deep = PAIR && (db.align_flags & UGS_A_DEEP) != 0;
    bool cont = false; uint32_t widx = 0;
    if constexpr (PAIR) if (bv.walk_units) { if (unit >= bv.n_walk) break; widx = unit; unit = bv.walk_units[widx]; cont = true; }
    uint32_t strands_left = 1, t_hits = 0;
    float t_min = 1.0f, t_max = 0.0f;                          // HitMgr::GetMinFractId / GetMaxFractId hitmgr.cpp:508-532
    if constexpr (PAIR) if (tflags) {
      if (unit >= bv.nq) break;
      unit *= bv.nstrand; strands_left = bv.nstrand;
    }
    if (unit >= units) break;
  next_strand:
    tq = ACLK();
    uint32_t qi = unit / bv.nstrand, strand = unit % bv.nstrand;
    if constexpr (PAIR) if (bv.unit_map) { const uint32_t m = bv.unit_map[unit]; qi = m >> 1; strand = m & 1u; }
    const uint64_t qo = bv.qoffs[qi];
    const uint32_t LA = (uint32_t)(bv.qoffs[qi + 1] - qo);
    c.LA = LA;
    uint32_t ncand = bv.cand_n[unit];
    uint32_t nvis = 0;
    bool walk_ended = false;          // the terminator ended the walk (not the end of the list)
    uint32_t nacc = 0, nrej = 0;
    // continuation of a parked walk: its counters, the page of its sorted key list that comes next, its overflow hit blocks
    uint32_t page = 0, n_total = 0, xhead = 0xffffffffu, xcur = 0xffffffffu;
    const uint64_t *dkeys = nullptr;
    bool setup_done = false;
    if constexpr (PAIR) if (cont) {
      const UgsWalkState st = bv.walk_state[unit];
      nacc = st.nacc; nrej = st.nrej; nvis = st.nvis; page = st.nvis;        // (a parked walk has no overflow block yet: its first pass saw at most K candidates)
      dkeys = bv.deep_keys + bv.deep_off[widx];
      n_total = (uint32_t)(bv.deep_off[widx + 1] - bv.deep_off[widx]);
      ncand = page < n_total ? (n_total - page < 64u ? n_total - page : 64u) : 0u;
    }
  next_page:
    // (the candidates' offsets and the first target's letters are asked for BEFORE the query set-up: three dependent trips to HBM that
    // used to stand between the set-up and the first pair now travel behind it)
    // one lane per candidate: id, offset and length fetched once for the whole unit
    uint32_t ct = 0, clen = 0; uint64_t cto = 0;
    if ((uint32_t)lane < ncand) {
      ct = bv.cand[(uint64_t)unit * K + lane];
      if constexpr (PAIR) if (cont) ct = (uint32_t)dkeys[page + (uint32_t)lane];        // (key = count | first-touch position: the low word is the target)
      cto = db.offs[ct];
      clen = (uint32_t)(db.offs[ct + 1] - cto);
    }
    // the NEXT candidate's letters travel while the current pair is aligned.  nt databases keep their letters a second time packed
    // 2 bits each (UgsDbView::pk, BASELINE north_star): a target of up to 1024 letters is then fetched as <= 65 uint2 (one stream: the 2-bit word and the "other letter" word of 16 letters side by side)
    // (lane j takes the words that hold its 16 letters) for the seed search and the ungapped extension, and its BYTES (case, IUPAC:
    // identity tests, DP scores) only if the pair gets as far as the chain gate - most pairs of a search are random candidates of
    // queries without a hit and never do.  Everything else (aa, longer targets, the in-batch pair stage of cluster_fast whose targets
    // are query letters): 4 x 4 letters per lane from the byte array as before.
    uint32_t pre[4] = {0, 0, 0, 0};
    const bool have_packed = NT && db.pk != nullptr;
    auto prefetch = [&](uint32_t k2) {
      const uint64_t to2 = ((uint64_t)(uint32_t)rl((int)(cto >> 32), (int)k2) << 32) | (uint32_t)rl((int)(uint32_t)cto, (int)k2);
      const uint32_t L2 = (uint32_t)rl((int)clen, (int)k2);
      if (have_packed && L2 <= 1024) {
        const uint64_t w0 = (to2 >> 4) + (uint32_t)lane;
        const bool on = (uint32_t)lane * 16u < L2 + 16u;                 // words 0 .. ceil(L2 / 16) (one more than the letters fill: the shift)
        uint2 lo = make_uint2(0u, 0u), hi = lo;
        if (on) { lo = db.pk[w0]; hi = db.pk[w0 + 1]; }
        pre[0] = lo.x; pre[1] = hi.x; pre[2] = lo.y; pre[3] = hi.y;
        return;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t o = (uint32_t)(e * 64 + lane) * 4;
        uint32_t v = 0;
        if (o < L2) __builtin_memcpy(&v, db.seqs + to2 + o, 4);     // unaligned dword load (the DB buffer is padded)
        pre[e] = v;
      }
    };
    if (ncand) prefetch(0);
    if (ncand && !setup_done) {
      setup_done = true;
      const int lane = fresh_lane();            // (a lane index of the set-up's own: see fresh_lane)
      // nt: the unit's letters arrive packed (2 bits + the "other letter" plane, k_rank_setup) - only the class bytes are made here
      bool planes = NT && bv.qpk != nullptr;
      if constexpr (PAIR) planes = planes && bv.unit_map == nullptr;
      for (uint32_t p = lane; p < LA; p += 64) {
        uint8_t ch = (strand == 0) ? bv.qseqs[qo + p] : s_comp[bv.qseqs[qo + (LA - 1 - p)]];
        const uint8_t cl = s_cls[ch];
        c.A[p] = cl;
        if (!planes) c.As[p] = s_sc[cl & 31];
      }
      if (planes) {
        const uint2 *qp = bv.qpk + (uint64_t)unit * bv.qpk_stride;
        const uint32_t nw = (LA + 15) >> 4;
        bool any = false;
        for (uint32_t k = lane; k < nw + 3; k += 64) { const uint2 e = qp[k]; c.A2[k] = e.x; c.Ai[k] = e.y; any = any || e.y != 0u; }
        if (lane < 2) { c.A2[-1 - lane] = 0; c.Ai[-1 - lane] = 0; }
        c.a_inv = __ballot(any) != 0;
      }
      wave_sync();
      if (NT && !planes) {
        pack_codes(c.As, LA, c.A2, c.Ai);
        bool any = false;
        for (uint32_t p = lane; p < LA; p += 64) any = any || c.As[p] > 3;
        c.a_inv = __ballot(any) != 0;
      }
      build_query_words(c, db.hsp_w, db.alpha);
    }
    ta0 += ACLK() - tq;
    // the group filter (group_filter above): once a unit has rejected db.group_after candidates, the next ones are tested UGS_GROUP at a time
    uint32_t pre_k = 0;                                         // the candidate whose letters `pre` holds
    uint32_t grp_lo = 0, grp_hi = 0, grp_maybe = 0;             // members [grp_lo, grp_hi) are decided: bit set = takes the full path
    uint32_t gwt = 0, gnB = 0, gmax = 0;
    bool group_ok = false;
    uint8_t *gtail = nullptr; uint32_t gcap = 0;                // (aa: the score codes of group members 1 .. 3, where the filter's seed list ends)
    if constexpr (!PAIR && !NT) {
      const uint32_t ub = (c.union_words * 4u) & ~15u;
      gmax = UGS_GROUP;
      if (ub >= 3u * maxt + 1024u) { gtail = (uint8_t *)c.seeds + (ub - 3u * maxt); gcap = (ub - 3u * maxt) / 4u; if (gcap > c.seed_cap) gcap = c.seed_cap; }
      group_ok = db.group_after != 0 && ncand && gtail != nullptr && c.use_tab && c.nwA > 0 && LA < 65536u;
    }
    if constexpr (!PAIR && NT) {
      gwt = (uint32_t)((((size_t)maxt / 16 + 6) * 4 + 15) & ~(size_t)15);
      gnB = maxt / gwt;
      gmax = gnB + (maxt + 32u) / gwt; if (gmax > UGS_GROUP) gmax = UGS_GROUP;
      group_ok = db.group_after != 0 && ncand && have_packed && c.use_tab && c.bsh == 0 && !c.a_inv && c.nwA > 0 && gmax >= 2 && LA < 65536u;
    }
    for (uint32_t k = 0; k < ncand; ++k) {
      tq = ACLK();
      if constexpr (!PAIR) {
        if (group_ok && k >= grp_hi && nrej >= db.group_after) {
          uint32_t n = ncand - k < gmax ? ncand - k : gmax;
          if (max_rej - nrej < n) n = max_rej - nrej;
          if (n >= 2) {
            uint32_t MinL = db.min_hsp_len_opt == 0 ? 32u : (uint32_t)db.min_hsp_len_opt;
            if (MinL > LA / 4) MinL = LA / 4;
            if (MinL < 16) MinL = 16;
            if constexpr (!NT) {
              GroupArgsAA ga;
              ga.seqs = db.seqs; ga.As = c.As; ga.Bs0 = c.Bs; ga.tail = gtail; ga.wstart = c.wstart; ga.qsort = c.qsort; ga.seeds = c.seeds;
              ga.s_cls = s_cls; ga.s_sc = s_sc; ga.s_hl = s_hl; ga.sub2 = s_sub2;
              ga.seed_cap = gcap; ga.LA = LA; ga.MinLength = MinL; ga.gstride = maxt; ga.bsh = c.bsh;
              ga.w = db.hsp_w; ga.alpha = db.alpha; ga.X = db.xdrop2; ga.minscore2 = db.minscore2;
              grp_maybe = uni(group_filter_aa(ga, k, n, cto, clen));
            } else {
            GroupArgs ga;
            ga.pk = db.pk; ga.A2 = c.A2; ga.B = c.B; ga.Bs = c.Bs; ga.wstart = c.wstart; ga.qsort = c.qsort; ga.seeds = c.seeds;
            ga.seed_cap = c.seed_cap; ga.LA = LA; ga.MinLength = MinL; ga.gwt = gwt; ga.nB = gnB;
            ga.w = db.hsp_w; ga.X = db.xdrop2; ga.m2 = c.s_sub2[0]; ga.mm2 = c.s_sub2[2]; ga.minscore2 = db.minscore2; ga.ctr = gclk; ga.xlut = c.xlut;
            grp_maybe = uni(group_filter(ga, k, n, cto, clen));
            }
            grp_lo = k; grp_hi = k + n;
            ta2 += ACLK() - tq; tq = ACLK();
          }
        }
        if (k < grp_hi && !((grp_maybe >> (k - grp_lo)) & 1u)) {
          // a pair without an HSP: fetched, counted and rejected as the full path would (no chain -> no alignment -> reject)
          nvis = page + k + 1;
          const uint32_t LBq = (uint32_t)rl((int)clen, (int)k);
          w_tletters += NT ? (((LBq + 15u) >> 4) + 1u) * 8u : LBq; ++w_pairs; ++w_grouped;
          ++nrej;
          if (nrej == max_rej) break;
          continue;
        }
        if (pre_k != k) { prefetch(k); pre_k = k; }
      }
      nvis = page + k + 1;
      const uint32_t t = (uint32_t)rl((int)ct, (int)k);
      const uint64_t to = ((uint64_t)(uint32_t)rl((int)(cto >> 32), (int)k) << 32) | (uint32_t)rl((int)(uint32_t)cto, (int)k);
      const uint32_t LB = (uint32_t)rl((int)clen, (int)k);
      c.LB = LB;
      const bool pack_direct = NT && LB <= 1024;             // nt: a lane's 4 letters are one byte of the packed arrays
      const bool from_packed = have_packed && LB <= 1024;
      bool classes_loaded = !from_packed;
      // the target's class bytes (c.B) for a pair fetched from the packed arrays: loaded when the pair first needs them
      auto load_classes = [&]() {
        if (classes_loaded) return;
        classes_loaded = true;
        w_tletters += LB;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t o = (uint32_t)(e * 64 + lane) * 4;
          if (o < LB) {
            uint32_t v = 0, cls4 = 0;
            __builtin_memcpy(&v, db.seqs + to + o, 4);
#pragma unroll
            for (uint32_t b = 0; b < 4; ++b) cls4 |= (uint32_t)s_cls[(v >> (8 * b)) & 0xffu] << (8 * b);
            *(uint32_t *)(c.B + o) = cls4;
          }
        }
        wave_sync();
      };
      bool anyb = false;                                       // a target letter that is not A/C/G/T/U
      if (from_packed) {
        const uint32_t nw = (LB + 15) >> 4, sh2 = ((uint32_t)to & 15u) * 2u;
        uint32_t w2 = __builtin_amdgcn_alignbit(pre[1], pre[0], sh2), wi = __builtin_amdgcn_alignbit(pre[3], pre[2], sh2);
        const uint32_t j = (uint32_t)lane;
        if (j + 1 == nw && (LB & 15u)) { const uint32_t m = (1u << (2u * (LB & 15u))) - 1u; w2 &= m; wi &= m; }   // letters behind the target's end belong to its neighbour
        if (j >= nw) { w2 = 0; wi = 0; }
        if (j < nw + 3) { c.B2[j] = w2; c.Bi[j] = wi; }
        if (nw + 3 > 64 && j < nw + 3 - 64) { c.B2[64 + j] = 0; c.Bi[64 + j] = 0; }
        if (lane < 2) { c.B2[-1 - lane] = 0; c.Bi[-1 - lane] = 0; }
        anyb = wi != 0;
      } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t o = (uint32_t)(e * 64 + lane) * 4;
        if (o < LB) {
          uint32_t p2 = 0, pi = 0, cls4 = 0;
#pragma unroll
          for (uint32_t b = 0; b < 4; ++b) {
            const uint8_t cl = s_cls[(pre[e] >> (8 * b)) & 0xffu];
            const uint32_t sc = s_sc[cl & 31];
            cls4 |= (uint32_t)cl << (8 * b);
            p2 |= (sc & 3u) << (2 * b); pi |= (sc >> 2) << (2 * b);
            anyb = anyb || (sc > 3u && o + b < LB);
            if (!pack_direct && o + b < LB) c.Bs[o + b] = (uint8_t)sc;
          }
          *(uint32_t *)(c.B + o) = cls4;                          // (classes behind the last letter are never read)
          if (pack_direct) { ((uint8_t *)c.B2)[o >> 2] = (uint8_t)p2; ((uint8_t *)c.Bi)[o >> 2] = (uint8_t)pi; }
        }
      }
      for (uint32_t p = 1024 + lane; p < LB; p += 64) { const uint8_t cl = s_cls[db.seqs[to + p]]; const uint8_t sc = s_sc[cl & 31]; c.B[p] = cl; c.Bs[p] = sc; anyb = anyb || sc > 3; }
      }
      c.b_inv = NT && __ballot(anyb) != 0;
      {
        // (no prefetch for a candidate the group filter has rejected already, or will look at itself)
        bool want = k + 1 < ncand;
        if constexpr (!PAIR) want = want && !(k + 1 < grp_hi ? !((grp_maybe >> (k + 1 - grp_lo)) & 1u) : (group_ok && nrej + 1 >= db.group_after && max_rej - nrej > 2 && ncand - k > 2));
        if (want) { prefetch(k + 1); pre_k = k + 1; }
      }
      wave_sync();
      if (NT && !pack_direct) { pack_codes(c.Bs, LB, c.B2, c.Bi); wave_sync(); }
      if constexpr (PAIR) if (db.pair_mask) {
        // Accepter::RejectPair accepter.cpp:140-197.  Big path: the pair is a reject for the terminator
        // (udbusortedsearcherbig.cpp:118-127); small path: it is passed over without a trace (searcher.cpp:63-67)
        const uint32_t pm = db.pair_mask;
        bool rej = false;
        if (pm & (UGS_P_SELF | UGS_P_NOTSELF)) {
          const bool same = bv.q_key[qi] == db.t_key[t];
          rej = ((pm & UGS_P_SELF) && same) || ((pm & UGS_P_NOTSELF) && !same);
        }
        if (!rej && (pm & UGS_P_SELFID) && LB == LA) {         // same length and identical stored letters
          load_classes();
          bool diff = false;
          for (uint32_t p0 = 0; p0 < LA && !diff; p0 += 64) { const uint32_t p = p0 + lane; diff = __ballot(p < LA && c.A[p] != c.B[p]) != 0; }
          rej = !diff;
        }
        if (!rej && (pm & UGS_P_MIN_SIZERATIO)) rej = (double)db.t_size[t] / (double)bv.q_size[qi] < (double)db.min_sizeratio;
        if (!rej && (pm & (UGS_P_MINQT | UGS_P_MAXQT | UGS_P_MINSL | UGS_P_MAXSL))) {
          const double qt = (double)LA / (double)LB, sl = (double)(LA < LB ? LA : LB) / (double)(LA > LB ? LA : LB);
          rej = ((pm & UGS_P_MINQT) && qt < (double)db.minqt) || ((pm & UGS_P_MAXQT) && qt > (double)db.maxqt) ||
                ((pm & UGS_P_MINSL) && sl < (double)db.minsl) || ((pm & UGS_P_MAXSL) && sl > (double)db.maxsl);
        }
        if (rej) {
          wave_sync();
          if (!db.big) continue;
          if (tflags && t_hits && (((tflags & UGS_A_TERMID) && (double)t_min <= (double)db.termid) ||
                                   ((tflags & UGS_A_TERMIDD) && (double)(t_max - t_min) > (double)db.termidd))) { walk_ended = true; break; }
          ++nrej;
          if (nrej == max_rej) { walk_ended = true; break; }
          continue;
        }
      }
      w_tletters += from_packed ? (((LB + 15u) >> 4) + 1u) * 8u : LB; ++w_pairs;      // bytes of the target fetched for this pair (packed planes, or the letters)
      // ---- GlobalAlign_AllOpts (globalalignmem.cpp:129-236), FailIfNoHSPs = true
      uint32_t MinHSPLength = db.min_hsp_len_opt == 0 ? 32u : (uint32_t)db.min_hsp_len_opt;
      if (MinHSPLength > LA / 4) MinHSPLength = LA / 4;
      if (MinHSPLength < 16) MinHSPLength = 16;
      ta1 += ACLK() - tq; tq = ACLK();
      // -fulldp: no HSPs at all, the whole pair is one unbanded hole (globalalignmem.cpp:148-152); -gaforce: FailIfNoHSPs = false
      const bool fulldp = PAIR ? (db.align_flags & UGS_A_FULLDP) != 0 : false;
      const bool force_all = PAIR ? (db.align_flags & (UGS_A_FULLDP | UGS_A_GAFORCE)) != 0 : false;
      if (!fulldp) {
        ungapped_blast<NT>(c, db, MinHSPLength, ctr);
      }
      ta2 += ACLK() - tq; tq = ACLK();
      unsigned long long tq2 = ACLK2();
      if (lane == 0) { if (fulldp) c.ws->nchain = 0; else chain_lane0(c); }
      wave_sync();
      ta0 += ACLK2() - tq2; tq2 = ACLK2();
      const uint32_t nchain = (uint32_t)__builtin_amdgcn_readfirstlane((int)c.ws->nchain);      // (written by lane 0, read by all: wave-uniform)
      bool accept = false;
      if (nchain || force_all) {
        load_classes();
        uint32_t TotLen = 0, TotSame = 0;
        for (uint32_t q = 0; q < nchain; ++q) {
          const HSPd h = uni(c.hsps[uni(c.chain[q])]);
          TotLen += h.Len;
          for (uint32_t x0 = 0; x0 < h.Len; x0 += 64) {
            const uint32_t x = x0 + lane;
            const bool idn = x < h.Len && ident(c, c.A[h.Loi + x], c.B[h.Loj + x]);
            TotSame += __popcll(__ballot(idn));
          }
        }
        const float HSPFractId = TotLen == 0 ? 0.0f : (float)TotSame / (float)TotLen;
        ta1 += ACLK2() - tq2; tq2 = ACLK2();
        if (force_all || !(HSPFractId < db.min_hsp_fract_id)) {
          // ---- stitch holes and HSPs into the path
          if (lane == 0) { c.ws->nruns = 0; c.ws->cur_len = 0; c.ws->cur_op = 0; c.ws->overflow = 0; }
          wave_sync();
          uint32_t pLoi = 0, pLoj = 0;                     // end (exclusive) of the previous HSP
          for (uint32_t q = 0; q < nchain; ++q) {
            const HSPd h = uni(c.hsps[uni(c.chain[q])]);
            align_hole(c, db, pLoi, pLoj, h.Loi - pLoi, h.Loj - pLoj, ctr);
            if (lane == 0) push_run(c, 0, h.Len);
            pLoi = h.Loi + h.Len; pLoj = h.Loj + h.Len;
          }
          align_hole(c, db, pLoi, pLoj, LA - pLoi, LB - pLoj, ctr);
          if (lane == 0) { flush_runs(c); if (c.ws->overflow) atomicOr(&ctr[UGS_CTR_ERR], (unsigned long long)UGS_ERR_RUNS); }
          wave_sync();
          const uint32_t nruns_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)c.ws->nruns);
          const uint32_t nr = nruns_w < c.runs_cap ? nruns_w : c.runs_cap;
          ta2 += ACLK2() - tq2; tq2 = ACLK2();
          // ---- AlignResult::FillLo on the run list
          int fm = -1, lm = -1; uint32_t cols = 0;
          for (uint32_t r = 0; r < nr; ++r) { const uint32_t run = uni(get_run(c, r)); cols += run >> 2; if ((run & 3) == 0) { if (fm < 0) fm = (int)r; lm = (int)r; } }
          if (fm >= 0) {
            uint32_t qpos = 0, tpos = 0, ids = 0, alen = 0, gaps = 0, opens = 0, mcols = 0, qlo = 0, tlo = 0, qhi = 0, thi = 0;
            uint32_t lastop = 0;
            for (uint32_t r = 0; r < nr; ++r) {
              const uint32_t run = uni(get_run(c, r)), op = run & 3, len = run >> 2;
              const bool inside = (int)r >= fm && (int)r <= lm;
              if ((int)r == fm) { qlo = qpos; tlo = tpos; }
              if (op == 0) {
                for (uint32_t x0 = 0; x0 < len; x0 += 64) {
                  const uint32_t x = x0 + lane;
                  const bool idn = x < len && ident(c, c.A[qpos + x], c.B[tpos + x]);
                  ids += __popcll(__ballot(idn));
                }
                mcols += len; qpos += len; tpos += len;
              } else if (op == 1) qpos += len; else tpos += len;
              if (inside) {
                alen += len;
                if (op != 0) { gaps += len; if (lastop == 0) ++opens; }
                lastop = op;
              }
              if ((int)r == lm) { qhi = qpos - 1; thi = tpos - 1; }
            }
            // Accepter::IsAcceptLo: FractId = double(ids)/double(alen) vs (double)(float)id
            accept = true;
            if (db.id_set) {
              const double FractId = alen == 0 ? 0.0 : (double)ids / (double)alen;
              if (FractId < db.id_accept) accept = false;
              if ((db.filter_mask & UGS_F_MAXID) && FractId > (double)db.maxid) accept = false;
            }
            if (db.filter_mask) {                          // the other optional filters of IsAcceptLo (accepter.cpp:41-91)
              const uint32_t fm = db.filter_mask, diffs = (mcols - ids) + gaps;
              if ((fm & UGS_F_MINCOLS) && alen < db.mincols) accept = false;
              if ((fm & UGS_F_MAXGAPS) && gaps > db.maxgaps) accept = false;
              if (fm & (UGS_F_QUERY_COV | UGS_F_MAX_QUERY_COV)) {
                const double Cov = (double)(qhi - qlo + 1) / (double)LA;           // arscorer.cpp:122-137
                if ((fm & UGS_F_QUERY_COV) && Cov < (double)db.query_cov) accept = false;
                if ((fm & UGS_F_MAX_QUERY_COV) && Cov > (double)db.max_query_cov) accept = false;
              }
              if (fm & (UGS_F_TARGET_COV | UGS_F_MAX_TARGET_COV)) {
                const double Cov = (double)mcols / (double)LB;                      // arscorer.cpp:139-154
                if ((fm & UGS_F_TARGET_COV) && Cov < (double)db.target_cov) accept = false;
                if ((fm & UGS_F_MAX_TARGET_COV) && Cov > (double)db.max_target_cov) accept = false;
              }
              if ((fm & UGS_F_MAXDIFFS) && diffs > db.maxdiffs) accept = false;
              if ((fm & UGS_F_MINDIFFS) && diffs < db.mindiffs) accept = false;
              if constexpr (PAIR) if ((fm & UGS_F_ABSKEW) && (double)db.t_size[t] / (double)bv.q_size[qi] < (double)db.abskew) accept = false;   // arscorer.cpp:809-816
            }
            if constexpr (PAIR) if (accept && tflags) {
              const float f = (float)(alen == 0 ? 0.0 : (double)ids / (double)alen);
              ++t_hits; t_min = f < t_min ? f : t_min; t_max = f > t_max ? f : t_max;
            }
            if (accept) {
              unsigned long long coff = 0;
              if (lane == 0) coff = atomicAdd(bv.cigar_used, (unsigned long long)nr);
              coff = ((unsigned long long)(uint32_t)rl((int)(coff >> 32), 0) << 32) | (uint32_t)rl((int)(uint32_t)coff, 0);
              if (coff + nr <= bv.cigar_cap)
                for (uint32_t r = lane; r < nr; r += 64) bv.cigar_pool[coff + r] = get_run(c, r);
              bool hslot = true;
              if constexpr (PAIR) if (nacc >= max_acc) {
                // beyond the unit's slots (deep walks only): blocks of UGS_XBLOCK hits chained per unit.  A pool that has run out is an
                // error flag and a demand (the counter keeps counting): the host grows the pool and runs the pass again
                if ((nacc - max_acc) % UGS_XBLOCK == 0u) {
                  uint32_t nb = 0;
                  if (lane == 0) nb = (uint32_t)atomicAdd(bv.xblocks_used, 1ull);
                  nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
                  if (nb < bv.xblocks_cap) {
                    if (lane == 0) { bv.xnext[nb] = 0xffffffffu; if (xcur != 0xffffffffu) bv.xnext[xcur] = nb; }
                    if (xcur == 0xffffffffu) xhead = nb;
                    xcur = nb;
                  } else { if (lane == 0) atomicOr(&ctr[UGS_CTR_ERR], (unsigned long long)UGS_ERR_XHITS); xcur = 0xfffffffeu; }
                }
                hslot = xcur < 0xfffffffeu;
              }
              if (lane == 0 && hslot) {
                ugs_hit *h = &bv.hits[(uint64_t)unit * max_acc + (nacc < max_acc ? nacc : 0u)];
                if constexpr (PAIR) if (nacc >= max_acc) h = &bv.xpool[(uint64_t)xcur * UGS_XBLOCK + (nacc - max_acc) % UGS_XBLOCK];
                h->query = qi; h->target = t; h->ids = ids; h->mism = mcols - ids; h->gaps_int = gaps; h->aln_len = alen;
                h->opens = opens; h->qlo = qlo; h->qhi = qhi; h->tlo = tlo; h->thi = thi; h->ql = LA; h->tl = LB;
                h->strand = strand; h->cigar_off = coff; h->cigar_len = nr; h->cols = cols; h->raw_score = 0.0f; h->flags = 0;
                atomicAdd(&ctr[UGS_CTR_HITS], 1ull);
              }
            }
          }
        }
      }
      ta3 += ACLK() - tq; ta3 += ACLK2() - tq2;
      // Terminator::Terminate (terminator.cpp:64-100)
      if constexpr (PAIR) if (tflags && t_hits && (((tflags & UGS_A_TERMID) && (double)t_min <= (double)db.termid) ||
                                                    ((tflags & UGS_A_TERMIDD) && (double)(t_max - t_min) > (double)db.termidd))) {
        if (accept) ++nacc;
        walk_ended = true;
        break;
      }
      if (accept) ++nacc; else ++nrej;
      if (nacc == (deep ? (uint32_t)db.acc_limit : max_acc)) { walk_ended = true; break; }
      if (nrej == max_rej) { if constexpr (PAIR) { if (!(db.align_flags & UGS_A_NOTERM)) { walk_ended = true; break; } } else break; }
      wave_sync();
    }
    // small path with pair filters: passed-over pairs do not count, so the walk may want more candidates than were kept
    if constexpr (PAIR) if (!deep && (db.pair_mask & UGS_P_SELFID) && !db.big && nacc < max_acc && nrej < max_rej && ncand == K && lane == 0) atomicOr(&ctr[UGS_CTR_ERR], (unsigned long long)UGS_ERR_PAIRCAP);      // (with deep walks such a walk is parked below)
    if constexpr (PAIR) if (deep) {
      if (cont) {
        // the next page of the unit's list
        if (!walk_ended && page + ncand < n_total) { page += ncand; ncand = n_total - page < 64u ? n_total - page : 64u; wave_sync(); goto next_page; }
        if (lane == 0) bv.walk_state[unit].xhead = xhead;       // (only this: the parked counters stay, the pass can be run again - ugs_host.cpp deep_stage)
      } else if (!walk_ended && nvis == ncand && ncand == K && lane == 0) {
        // a walk that ran through a FULL candidate list without meeting a limit has more to visit: parked for the continuation pass
        UgsWalkState st; st.nacc = nacc; st.nrej = nrej; st.nvis = nvis; st.xhead = 0xffffffffu; st.xcur = 0xffffffffu; st.pad0 = st.pad1 = st.pad2 = 0;
        bv.walk_state[unit] = st;
        bv.open_list[atomicAdd(&ctr[UGS_CTR_OPEN], 1ull)] = unit;
      }
    }
    if (lane == 0) { bv.hit_n[unit] = nacc; if (bv.walk_n) bv.walk_n[unit] = nvis; }
    wave_sync();
    if constexpr (PAIR) if (--strands_left) { ++unit; goto next_strand; }
  }
  if (lane == 0) { atomicAdd(&ctr[UGS_CTR_TLETTERS], w_tletters); atomicAdd(&ctr[UGS_CTR_PAIRS], w_pairs); if (w_grouped) atomicAdd(&ctr[UGS_CTR_GROUPED], w_grouped); }
  if (tid == 0) {
#if UGS_ALIGN_CLOCKS == 5
    atomicAdd(&ctr[UGS_CTR_T4], gclk[0]); atomicAdd(&ctr[UGS_CTR_T5], gclk[1]); atomicAdd(&ctr[UGS_CTR_T6], gclk[2]); atomicAdd(&ctr[UGS_CTR_T7], gclk[3]);
#elif UGS_ALIGN_CLOCKS != 3
    atomicAdd(&ctr[UGS_CTR_T4], ta0); atomicAdd(&ctr[UGS_CTR_T5], ta1); atomicAdd(&ctr[UGS_CTR_T6], ta2); atomicAdd(&ctr[UGS_CTR_T7], ta3);
#endif
  }
}

static const void *align_kernel(bool pair, bool nt)
{
  return pair ? (nt ? (const void *)k_align<true, true> : (const void *)k_align<true, false>) : (nt ? (const void *)k_align<false, true> : (const void *)k_align<false, false>);
}

int ugs_align_blocks_per_cu(int threads, size_t lds, int is_nucleo)
{
  int n = 0;
  const void *fn = align_kernel(false, is_nucleo != 0);
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, threads, lds) != hipSuccess || n < 1) n = 1;
  return n;
}

int ugs_launch_align(const UgsDbView &db, const UgsBatchView &b, const UgsAlignLaunch &L, hipStream_t st)
{
  uint32_t wave_lds = (uint32_t)((L.lds - UGS_ALIGN_HDR) / L.wpb);
  const void *fn = align_kernel(db.pair_mask || (db.filter_mask & UGS_F_ABSKEW) || db.align_flags, db.is_nucleo != 0);
  HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds));
  UgsDbView a0 = db; UgsBatchView a1 = b; uint32_t a2 = L.hsp_cap, a4 = L.seed_cap;
  void *args[] = {&a0, &a1, &a2, &wave_lds, &a4};
  if (ugs_kernel_log) ugs_before_launch("k_align");
  HIPCHK(hipLaunchKernel(fn, dim3(L.grid), dim3(64 * L.wpb), args, L.lds, st));
  if (ugs_kernel_log) ugs_after_launch("k_align", st);
  HIPCHK(hipGetLastError());
  return UGS_OK;
}
