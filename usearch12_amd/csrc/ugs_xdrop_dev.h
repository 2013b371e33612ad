// ugs_xdrop_dev.h - device side of the gapped x-drop extension, shared by the batched
// kernel (ugs_xdrop.hip, ugs_xdrop_batch) and the usearch_local driver kernel (ugs_local.hip).
// One wavefront per extension, lanes = columns of the live window; see ugs_xdrop.hip for the mapping.
#pragma once
#include "ugs_dev.h"

namespace {


constexpr int NEG = -(1 << 29);
constexpr uint32_t XD_MAXL = 4096;            // xdpmem.h:6 g_MaxL
constexpr uint8_t TB_DM = 1, TB_IM = 2, TB_MD = 4, TB_MI = 8;   // tracebit.h:4-7
enum { ST_OK = 0, ST_TB = 1, ST_ARENA = 2, ST_BAD = 3 };

struct XdView {
  const uint8_t *a_seqs; const uint64_t *a_offs;
  const uint8_t *b_seqs; const uint64_t *b_offs;
  const ugs_xdrop_job *jobs; uint32_t njobs;
  const uint32_t *job_list;                   // retry pass: indices of the jobs to run (nullptr = all)
  ugs_xdrop_hsp *hsps;
  uint32_t *status;
  uint32_t *arena; unsigned long long arena_cap; unsigned long long *arena_used;
  uint8_t *tb; unsigned long long tb_cap;     // per wave
  uint2 *rowinfo; uint32_t rows_cap;          // per wave: (offset into tb, jlo | width<<16)
  uint32_t *runbuf; uint32_t runbuf_cap;      // per wave: two halves (forward side, backward side)
  unsigned int *next_job; unsigned long long *cells;
  const int8_t *sub2; const uint8_t *cls;
  int open2, ext2; float X, abs_open, abs_ext;
  uint32_t W;                                 // LDS row width in ints
};

struct XdWave {
  int lane;
  int *M0, *M1, *D; uint8_t *Bc;
  const int8_t *sub2; const uint8_t *cls;     // LDS copies
  uint8_t *tb; uint2 *rowinfo;
};

// Cross-lane traffic through HBM scratch (traceback bytes, row windows, run lists: one lane stores, another lane of
// the SAME wave loads).  UGS_XD_SYNC=1 (product): the writer side waits for its stores to be acknowledged by L2
// (s_waitcnt vmcnt(0) - a workgroup-scope fence emits no such wait on gfx950, it trusts the CU's L1 to be coherent
// with the CU's own stores) and the reader side uses agent-scope loads (sc1: served by L2, never by a line the
// CU's L1 still holds from an earlier job).  UGS_XD_SYNC=0 is the round-1/2 behaviour, kept for A/B runs only.
#ifndef UGS_XD_SYNC
#define UGS_XD_SYNC 1
#endif
__device__ __forceinline__ void lds_sync() { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
#if UGS_XD_SYNC
__device__ __forceinline__ void wave_sync() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ uint8_t xl8(const uint8_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t xl32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint2 xl64(const uint2 *p)
{
  const unsigned long long x = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_uint2((uint32_t)x, (uint32_t)(x >> 32));
}
#else
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ uint8_t xl8(const uint8_t *p) { return *p; }
__device__ __forceinline__ uint32_t xl32(const uint32_t *p) { return *p; }
__device__ __forceinline__ uint2 xl64(const uint2 *p) { return *p; }
#endif
__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// inclusive prefix max over the 64 lanes (DPP inside 16-lane rows, then the three row totals)
__device__ __forceinline__ int wave_incl_max(int v)
{
  int x;
  x = __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false); v = imax(v, x);   // row_shr:1
  x = __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false); v = imax(v, x);   // row_shr:2
  x = __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false); v = imax(v, x);   // row_shr:4
  x = __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false); v = imax(v, x);   // row_shr:8
  const int t0 = rl(v, 15), t1 = imax(t0, rl(v, 31)), t2 = imax(t1, rl(v, 47));
  const int row = (int)(threadIdx.x & 63) >> 4;
  const int add = row == 1 ? t0 : (row == 2 ? t1 : t2);
  return row == 0 ? v : imax(v, add);
}
__device__ __forceinline__ int wave_max(int v) { return rl(wave_incl_max(v), 63); }
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) { return ~(uint32_t)wave_max((int)(~v ^ 0x80000000u)) ^ 0x80000000u; }
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t u)
{
  int v = (int)u, x;
  x = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); v += x;
  x = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); v += x;
  x = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); v += x;
  x = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); v += x;
  const int t0 = rl(v, 15), t1 = t0 + rl(v, 31), t2 = t1 + rl(v, 47);
  const int row = (int)(threadIdx.x & 63) >> 4;
  return (uint32_t)(v + (row == 0 ? 0 : (row == 1 ? t0 : (row == 2 ? t1 : t2))));
}
// value of lane-1 (lane 0 gets `first`)
__device__ __forceinline__ int shift_up1(int v, int first, int lane)
{
  const int x = __shfl_up(v, 1);
  return lane == 0 ? first : x;
}
__device__ __forceinline__ int hibit(uint64_t m) { return 63 - __clzll((long long)m); }

// one traceback byte: TB[r][c] of the reference's full matrix, from the per-row windows
__device__ __forceinline__ uint8_t tbget(const XdWave &w, uint32_t lastrow, uint32_t r, uint32_t c)
{
  if (r < 1 || r > lastrow) return 0;
  const uint2 ri = xl64(&w.rowinfo[r]);
  const uint32_t jlo = ri.y & 0xffffu, width = ri.y >> 16;
  if (c < jlo || c - jlo >= width) return 0;
  return xl8(&w.tb[ri.x + (c - jlo)]);
}

// XDropFwdFastMem on A[0], A[sa], ... / B[0], B[sb], ...  (sa = sb = -1 walks backwards).
// Appends the path as runs in OUTWARD order (origin -> best cell) to runs[nruns..].
// Returns the score in half-units.
__device__ __forceinline__ int xd_extend(const XdView &v, XdWave &w, const uint8_t *A, int sa, uint32_t LA,
                                         const uint8_t *B, int sb, uint32_t LB, uint32_t &Leni, uint32_t &Lenj,
                                         uint32_t *runs, uint32_t &nruns, bool &ovf, unsigned long long &cells)
{
  const int lane = w.lane;
  const int s00 = w.sub2[((int)w.cls[A[0]] << 5) | w.cls[B[0]]];
  if (LA == 1 || LB == 1) {                                  // xdropfwdmem.cpp:360-368
    Leni = 1; Lenj = 1;
    if (lane == 0 && nruns < v.runbuf_cap) runs[nruns] = (1u << 2);
    ++nruns;
    wave_sync();
    return s00;
  }
#if defined(UGS_XD_POISON)
  // debugging build: whatever this call reads without having written it shows up as a wrong answer
  for (unsigned long long k = lane; k < v.tb_cap && k < (2ull << 20); k += 64) w.tb[k] = 0xff;
  for (uint32_t k = lane; k < LA + 1; k += 64) w.rowinfo[k] = make_uint2(0xffffffffu, 0xffffffffu);
  for (uint32_t k = lane; k < LB + 4; k += 64) { w.M0[k] = 0x3f3f3f3f; w.M1[k] = 0x3f3f3f3f; w.D[k] = 0x3f3f3f3f; w.Bc[k] = 0x1f; }
  wave_sync();
#endif
  for (uint32_t j = lane; j < LB; j += 64) w.Bc[j] = w.cls[B[(long)j * sb]];
  int *Mcur = w.M0, *Mnxt = w.M1, *D = w.D;
  if (lane == 0) { Mcur[1] = s00; D[1] = NEG; }
  lds_sync();
  const int open2 = v.open2, ext2 = v.ext2;
  const float X = v.X, absOpen = v.abs_open, absExt = v.abs_ext;
  int Best = s00;
  uint32_t Besti = 0, Bestj = 0;
  uint32_t vlo = 1, vhi = 1, jlo = 1, jhi = 1, prev_jhi = 0;
  uint32_t tboff = 0, lastrow = 0;
  int areg = 0;
  for (uint32_t i = 1; i < LA; ++i) {
    if (((i - 1) & 63) == 0) { const uint32_t ii = i + lane; areg = ii < LA ? w.cls[A[(long)ii * sa]] : 0; }
    const int a = rl(areg, (int)((i - 1) & 63));
    const int8_t *subrow = w.sub2 + (a << 5);
    const uint32_t jhi_init = jhi;
    const bool quirk_ok = (jhi_init == prev_jhi + 1);        // endj == jhi, see the I-part note below
    uint32_t next_jlo = 0xffffffffu, next_jhi = 0xffffffffu;
    int carryI = NEG;
    uint32_t jend = 0;
    uint32_t c0 = jlo;
    bool donly = false;                                      // row end fell on lane 63: one more pass for column jend+1
    uint8_t *tbrow = w.tb + tboff;
    for (;;) {
      if ((unsigned long long)tboff + (c0 - jlo) + 66 > v.tb_cap) { ovf = true; return 0; }
      const uint32_t j = c0 + lane;
      const bool in = (j >= vlo) & (j <= vhi);
      const int mp = in ? Mcur[j] : NEG;                     // DPM[i][j]
      const int dp = in ? D[j] : NEG;                        // DPD[i][j]
      const int md = mp + open2;
      int dn = dp + ext2;
      const bool mdbit = md >= dn;
      if (mdbit) dn = md;                                    // DPD[i+1][j]
      if (donly) {                                           // "end of Drow" xdropfwdmem.cpp:645-666
        if (lane == 0) { D[j] = dn; tbrow[j - jlo] = mdbit ? TB_MD : 0; }
        break;
      }
      const int sub = j < LB ? subrow[w.Bc[j]] : 0;
      // insert chain I[j+1] = max(M[j] + open, I[j] + ext) as a max-plus prefix scan
      int sc = wave_incl_max(md - lane * ext2);
      const int fromscan = sc + lane * ext2;
      const int fromcarry = carryI + (lane + 1) * ext2;
      const int Iout = imax(fromscan, fromcarry);            // DPI[i][j+1]
      const int Iin = shift_up1(Iout, carryI, lane);         // DPI[i][j]
      uint8_t bits = 0;
      int xM = mp;
      if (dp > xM) { xM = dp; bits = TB_DM; }
      if (Iin > xM) { xM = Iin; bits = TB_IM; }
      const int s = xM + sub;                                // DPM[i+1][j+1]
      if (md >= Iin + ext2) bits |= TB_MI;
      if (j != jlo && mdbit) bits |= TB_MD;
      // running best: cell j tests against the best of all earlier cells (row-major)
      const int pm = wave_incl_max(s);
      const int Bm = imax(Best, shift_up1(pm, Best, lane));
      const int Bdi = imax(Bm, s);
      const float hM = (float)(s - Bm) * 0.5f + X;
      const float hD = (float)(dn - Bdi) * 0.5f + X;
      const float hI = (float)(Iout - Bdi) * 0.5f + X;
      const bool EM = hM > absExt, EI = hI > absExt;
      const bool E = (EM | EI) & (j + 1 < LB);               // this cell, if last, extends the row
      const uint64_t Emask = __ballot(E);
      int e = 63;
      bool endhere = false;
      if (jhi < c0 + 64) {
        const int t = jhi > c0 ? (int)(jhi - c0) : 0;
        const uint64_t notE = ~Emask & (~0ull << t);
        if (notE) { e = __ffsll((long long)notE) - 1; endhere = true; }
      }
      const bool act = lane <= e;
      // the reference's I-part extension re-initialises Mrow[endj..] without sparing the value the
      // current cell just stored (the M-part does spare it, xdropfwdmem.cpp:538-546 vs :625-631):
      // when the row's first extension comes from the insert test alone and endj == j, DPM[i+1][j+1]
      // is overwritten with -inf.  Results depend on it, so it is reproduced.
      const bool clob = quirk_ok & (j == jhi_init) & !EM & EI & (j + 1 < LB);
      // best cell: '>=' so the last cell of the maximum wins (:555)
      const int mx = wave_max(act ? s : (int)0x80000000);
      if (mx >= Best) { Best = mx; Besti = i; Bestj = c0 + (uint32_t)hibit(__ballot(act && s == mx)); }
      const bool mm = act && hM > 0.0f, im = act && hI > 0.0f, dmv = act && j != jlo && hD > 0.0f;
      uint32_t cand = 0xffffffffu;
      if (mm | im) cand = j + 1;
      if (act && hM > absOpen) cand = j;
      if (dmv) cand = j - 1;
      next_jlo = min(next_jlo, wave_min_u32(cand));
      const uint64_t Am = __ballot(mm | im), Dm = __ballot(dmv);
      if (Am) {                                              // next_jhi: the last assignment wins, later Delete-Match may raise it
        const int la = hibit(Am);
        uint32_t nh = c0 + (uint32_t)la + 1;
        const uint64_t above = la == 63 ? 0ull : (Dm & (~0ull << (la + 1)));
        if (above) nh = max(nh, c0 + (uint32_t)hibit(above) - 1);
        next_jhi = nh;
      } else if (Dm) next_jhi = max(next_jhi, c0 + (uint32_t)hibit(Dm) - 1);
      if (act) {
        Mnxt[j + 1] = clob ? NEG : s;
        if (j != jlo) D[j] = dn;
        tbrow[j - jlo] = bits;
      }
      if (endhere) {
        jend = c0 + (uint32_t)e;
        if (e < 63) {
          if (lane == e + 1) { D[j] = dn; tbrow[j - jlo] = mdbit ? TB_MD : 0; }
          break;
        }
        donly = true;
      } else {
        carryI = rl(Iout, 63);
        if (jhi < c0 + 64) jhi = c0 + 64;
      }
      c0 += 64;
    }
    if (lane == 0) w.rowinfo[i] = make_uint2(tboff, jlo | ((jend - jlo + 2) << 16));
    tboff += jend - jlo + 2;
    cells += jend - jlo + 1;
    lastrow = i;
    if (next_jlo == 0xffffffffu) break;
    vlo = jlo + 1; vhi = jend + 1; prev_jhi = jend;
    jlo = min(next_jlo, LB - 1); jhi = min(next_jhi, LB - 1);
    int *t = Mcur; Mcur = Mnxt; Mnxt = t;
    lds_sync();
  }
  if (Best <= 0) { Leni = 0; Lenj = 0; return 0; }           // :712-718
  wave_sync();                                               // traceback bytes + rowinfo visible to every lane
  // XDropFwdTraceBackBitMem :271-342 (every lane walks the same cells; loads are wave-uniform)
  const uint32_t n0 = nruns;
  {
    uint32_t i = Besti, j = Bestj, st = 0, curop = 0, curlen = 0;
    const uint32_t guard = LA + LB + 4;
    for (uint32_t step = 0; step < guard; ++step) {
      if (st == curop) ++curlen;
      else {
        if (lane == 0 && nruns < v.runbuf_cap) runs[nruns] = (curlen << 2) | curop;
        ++nruns; curop = st; curlen = 1;
      }
      if (i == 0 && j == 0) break;
      if (st == 0) { const uint8_t t = tbget(w, lastrow, i, j); st = (t & TB_DM) ? 1 : ((t & TB_IM) ? 2 : 0); --i; --j; }
      else if (st == 1) { const uint8_t t = tbget(w, lastrow, i, j + 1); st = (t & TB_MD) ? 0 : 1; --i; }
      else { const uint8_t t = tbget(w, lastrow, i + 1, j); st = (t & TB_MI) ? 0 : 2; --j; }
      if ((int)i < 0 || (int)j < 0) break;                   // cannot happen on a consistent matrix
    }
    if (lane == 0 && nruns < v.runbuf_cap) runs[nruns] = (curlen << 2) | curop;
    ++nruns;
  }
  wave_sync();
  {                                                          // traceback order is inward; store outward
    const uint32_t n = nruns - n0;
    for (uint32_t k = lane; k < n / 2; k += 64) {
      const uint32_t x = xl32(&runs[n0 + k]), y = xl32(&runs[n0 + n - 1 - k]);
      runs[n0 + k] = y; runs[n0 + n - 1 - k] = x;
    }
  }
  wave_sync();
  Leni = Besti + 1; Lenj = Bestj + 1;
  return Best;
}

__device__ __forceinline__ uint32_t xd_subl(uint32_t L)      // GetSubL xdropfwdsplit.cpp:15-22
{
  if (L <= XD_MAXL) return L;
  if (L < 2 * XD_MAXL) return L / 2;
  return XD_MAXL;
}

// one side of XDropAlignMem: plain call or the split driver (XDropFwdSplit / XDropBwdSplit are the
// same loop once "backwards" is a stride)
__device__ __forceinline__ int xd_side(const XdView &v, XdWave &w, const uint8_t *A, int sa, uint32_t LA,
                                       const uint8_t *B, int sb, uint32_t LB, bool split, uint32_t &Leni, uint32_t &Lenj,
                                       uint32_t *runs, uint32_t &nruns, bool &ovf, unsigned long long &cells)
{
  if (!split) return xd_extend(v, w, A, sa, LA, B, sb, LB, Leni, Lenj, runs, nruns, ovf, cells);
  Leni = 0; Lenj = 0;
  int sum = 0;
  for (;;) {
    if (Leni == LA || Lenj == LB) break;
    const uint32_t SubLA = xd_subl(LA - Leni), SubLB = xd_subl(LB - Lenj);
    uint32_t si = 0, sj = 0;
    const int sc = xd_extend(v, w, A + (long)Leni * sa, sa, SubLA, B + (long)Lenj * sb, sb, SubLB, si, sj, runs, nruns, ovf, cells);
    if (ovf) return 0;
    if (sc == 0) break;
    sum += sc; Leni += si; Lenj += sj;
    if (si < SubLA && sj < SubLB) break;
  }
  return sum;
}

}  // namespace
