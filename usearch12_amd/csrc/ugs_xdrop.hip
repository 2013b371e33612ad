// ugs_xdrop.hip - batched gapped x-drop extension for gfx950.
//
// What it replaces (reference, one call per anchor from LocalAligner, localaligner.cpp:190,:330):
//   XDropFwdFastMem  xdropfwdmem.cpp:344-749 (+ traceback :271-342)
//   XDropBwdFastMem  xdropbwdmem.cpp:23-70, XDropFwdSplit/BwdSplit xdropfwdsplit.cpp:24-91, xdropbwdsplit.cpp:15-79
//   XDropAlignMem    xdropalignmem.cpp:26-244
//
// Mapping: one wavefront per job, lanes = columns of the live window [jlo, jhi] of one DP row.
// The reference walks a row cell by cell with three running quantities - the insert score I0, the
// best score so far (which feeds every h = s - Best + X test of later cells in the SAME row) and
// the row end, which grows while the last cell still passes the extension test.  All three are
// prefix computations: I0 is a max-plus scan, the running best an exclusive prefix max, and the
// row end the first lane at or after the initial jhi whose extension test fails (cells beyond the
// end are computed speculatively and discarded).  next_jlo / next_jhi follow the reference's
// sequential min / "last assignment wins" rules, evaluated with ballots.  Scores are kept as
// int32 half-units (exact: the reference's floats only ever hold small multiples of 0.5).
// Rows live in LDS (two M rows ping-pong + one D row + the target's letter classes), the
// traceback bits in a per-wave HBM scratch, compacted per row to the live window.
//
// A backwards extension never copies reversed sequences: it walks the letters with stride -1.
#include "ugs_dev.h"
#include <rocprim/device/device_scan.hpp>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>

#include "ugs_xdrop_dev.h"

namespace {


__global__ __launch_bounds__(256) void k_xdrop(XdView v)
{
  extern __shared__ __align__(16) unsigned char smem[];
  int8_t *s_sub2 = (int8_t *)smem;
  uint8_t *s_cls = (uint8_t *)(smem + 1024);
  for (uint32_t k = threadIdx.x; k < 1024; k += blockDim.x) s_sub2[k] = v.sub2[k];
  for (uint32_t k = threadIdx.x; k < 256; k += blockDim.x) s_cls[k] = v.cls[k];
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t wpb = blockDim.x >> 6;
  const uint32_t gw = blockIdx.x * wpb + (uint32_t)wave;
  const size_t wave_lds = (size_t)v.W * 13;                  // 3 int rows + 1 byte row
  unsigned char *base = smem + 1280 + (size_t)wave * ((wave_lds + 15) & ~(size_t)15);
  XdWave w;
  w.lane = (int)(threadIdx.x & 63);
  w.M0 = (int *)base; w.M1 = w.M0 + v.W; w.D = w.M1 + v.W; w.Bc = (uint8_t *)(w.D + v.W);
  w.sub2 = s_sub2; w.cls = s_cls;
  w.tb = v.tb + (size_t)gw * v.tb_cap;
  w.rowinfo = v.rowinfo + (size_t)gw * v.rows_cap;
  uint32_t *runsF = v.runbuf + (size_t)gw * 2 * v.runbuf_cap, *runsB = runsF + v.runbuf_cap;
  const int lane = w.lane;
  unsigned long long cells = 0;
  const uint32_t total = v.njobs;
  for (;;) {
    uint32_t slot = 0;
    if (lane == 0) slot = atomicAdd(v.next_job, 1u);
    slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
    if (slot >= total) break;
    const uint32_t jid = v.job_list ? v.job_list[slot] : slot;
    const ugs_xdrop_job job = v.jobs[jid];
    const uint8_t *A = v.a_seqs + v.a_offs[job.a], *B = v.b_seqs + v.b_offs[job.b];
    const uint32_t LA = (uint32_t)(v.a_offs[job.a + 1] - v.a_offs[job.a]), LB = (uint32_t)(v.b_offs[job.b + 1] - v.b_offs[job.b]);
    uint32_t nf = 0, nb = 0, mid = 0;
    uint32_t loi = 0, loj = 0, leni = 0, lenj = 0;
    int score2 = 0;
    bool ovf = false, empty = false;
    if (job.mode == UGS_XDROP_FWD) {
      score2 = xd_extend(v, w, A, 1, LA, B, 1, LB, leni, lenj, runsF, nf, ovf, cells);
    } else if (job.mode == UGS_XDROP_BWD) {
      score2 = xd_extend(v, w, A + (LA - 1), -1, LA, B + (LB - 1), -1, LB, leni, lenj, runsB, nb, ovf, cells);
      loi = LA - leni; loj = LB - lenj;
    } else if (job.anc_len <= 1) {
      empty = true;                                          // xdropalignmem.cpp:44-49
    } else {
      const uint32_t AncLoi = job.anc_loi, AncLoj = job.anc_loj, AncLen = job.anc_len;
      const uint32_t AncHii = AncLoi + AncLen - 1, AncHij = AncLoj + AncLen - 1;
      uint32_t bi = 0, bj = 0, fi = 0, fj = 0;
      const int bwd = xd_side(v, w, A + AncLoi, -1, AncLoi + 1, B + AncLoj, -1, AncLoj + 1,
                              AncLoi > XD_MAXL || AncLoj > XD_MAXL, bi, bj, runsB, nb, ovf, cells);
      int fwd = 0;
      if (!ovf) fwd = xd_side(v, w, A + AncHii, 1, LA - AncHii, B + AncHij, 1, LB - AncHij,
                              (LA - AncHii) > XD_MAXL || (LB - AncHij) > XD_MAXL, fi, fj, runsF, nf, ovf, cells);
      int anc = 0;
      for (uint32_t k = lane; k < AncLen; k += 64) anc += s_sub2[((int)s_cls[A[AncLoi + k]] << 5) | s_cls[B[AncLoj + k]]];
      for (int o = 32; o; o >>= 1) anc += __shfl_xor(anc, o);
      const int dupe = s_sub2[((int)s_cls[A[AncLoi]] << 5) | s_cls[B[AncLoj]]] + s_sub2[((int)s_cls[A[AncHii]] << 5) | s_cls[B[AncHij]]];
      score2 = bwd + fwd + anc - dupe;                       // :176
      loi = AncLoi + 1 - bi; loj = AncLoj + 1 - bj;
      leni = bi + fi + AncLen - 2; lenj = bj + fj + AncLen - 2;
      mid = AncLen - 2;
    }
    uint32_t st = ST_OK;
    uint32_t nheads = 0;
    unsigned long long off = 0;
    if (ovf) st = ST_TB;
    else if (!empty) {
      if (nf > v.runbuf_cap || nb > v.runbuf_cap) st = ST_BAD;
      else {
        // path = reverse(backward side) + M x mid + forward side, adjacent equal ops merged
        const uint32_t nm = mid ? 1u : 0u, n = nb + nm + nf;
        auto elem = [&](uint32_t k) -> uint32_t {
          if (k < nb) return xl32(&runsB[nb - 1 - k]);
          if (k < nb + nm) return mid << 2;
          return xl32(&runsF[k - nb - nm]);
        };
        for (uint32_t k0 = 0; k0 < n; k0 += 64) {
          const uint32_t k = k0 + lane;
          const bool head = k < n && (k == 0 || ((elem(k) ^ elem(k - 1)) & 3u));
          nheads += (uint32_t)__popcll(__ballot(head));
        }
        if (lane == 0) off = atomicAdd(v.arena_used, (unsigned long long)nheads);
        off = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) |
              (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)off);
        if (off + nheads > v.arena_cap) st = ST_ARENA;
        else {
          // merged run = segment of equal ops; the LAST element of a segment writes it (one plain store per output
          // word, no read-modify-write): length = running sum at the tail - running sum before the segment's head,
          // the head found by ballot inside the chunk or carried in a scalar from the chunks before
          uint32_t *out = v.arena + off;
          uint32_t before = 0, sbase = 0, seg_start = 0;
          for (uint32_t k0 = 0; k0 < n; k0 += 64) {
            const uint32_t k = k0 + lane;
            const uint32_t r = k < n ? elem(k) : 0;
            const bool head = k < n && (k == 0 || ((r ^ elem(k - 1)) & 3u));
            const bool tail = k < n && (k + 1 == n || ((r ^ elem(k + 1)) & 3u));
            const uint32_t len = r >> 2;
            const uint32_t S = sbase + wave_incl_sum(len), Sx = S - len;
            const uint64_t hm = __ballot(head);
            const uint64_t below = hm & ((2ull << lane) - 1);
            const uint32_t sp = (uint32_t)__shfl((int)Sx, below ? hibit(below) : 0);
            const uint32_t start = below ? sp : seg_start;
            if (tail) out[before + (uint32_t)__popcll(below) - 1] = ((S - start) << 2) | (r & 3u);
            if (hm) seg_start = (uint32_t)rl((int)Sx, hibit(hm));
            sbase = (uint32_t)rl((int)S, 63);
            before += (uint32_t)__popcll(hm);
          }
        }
      }
    }
    if (lane == 0) {
      ugs_xdrop_hsp h;
      h.score = (st == ST_OK) ? (float)score2 * 0.5f : 0.0f;
      h.loi = loi; h.loj = loj; h.leni = leni; h.lenj = lenj;
      h.path_len = (st == ST_OK) ? nheads : 0; h.path_off = off;
      v.hsps[jid] = h;
      v.status[jid] = st;
    }
    wave_sync();
  }
  if (lane == 0 && cells) atomicAdd(v.cells, cells);
}

// arena (allocation order) -> pool in job order
__global__ void k_xd_compact(const uint32_t *arena, ugs_xdrop_hsp *hsps, const uint64_t *dst_off, uint32_t njobs, uint32_t *pool)
{
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= njobs) return;
  const ugs_xdrop_hsp h = hsps[wave];
  const uint64_t d = dst_off[wave];
  for (uint32_t k = lane; k < h.path_len; k += 64) pool[d + k] = arena[h.path_off + k];
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) hsps[wave].path_off = d;
}

__global__ void k_xd_lens(const ugs_xdrop_hsp *hsps, uint32_t njobs, uint64_t *len)
{
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < njobs) len[k] = hsps[k].path_len;
}

thread_local float g_last_ms = 0.0f;
thread_local uint64_t g_last_cells = 0;

// UGS_XD_HOSTMODE (A/B builds, tools/build_xd_variant.sh): which of the call's stream / two events are created and
// destroyed per call.  0 = all three per call (rounds 1-2), 1 = none (product: kept per host thread and device),
// 3 = stream per call, events kept, 4 = events per call, stream kept.
#ifndef UGS_XD_HOSTMODE
#define UGS_XD_HOSTMODE 1
#endif
// (kept until the process ends - like the index build's scratch, ugs_index.hip: the device may be gone when thread-locals are destroyed;
// a thread that uses more than eight devices falls back to a stream and events of its own per call)
struct XdHostCtx { int dev = -1; hipStream_t st = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr; };
thread_local XdHostCtx g_ctx[8];

struct DevBufs {
  std::vector<void *> ptrs;
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool own_st = false, own_ev = false;
  ~DevBufs()
  {
    for (void *p : ptrs) (void)ugs_free(p);
    if (own_ev && e0) (void)hipEventDestroy(e0);
    if (own_ev && e1) (void)hipEventDestroy(e1);
    if (own_st && st) (void)hipStreamDestroy(st);
  }
  hipError_t open(int device)
  {
    XdHostCtx *c = nullptr;
    for (auto &x : g_ctx) if (x.dev == device) c = &x;
    if (!c) for (auto &x : g_ctx) if (x.dev < 0) { c = &x; x.dev = device; break; }
    const bool keep_st = c && (UGS_XD_HOSTMODE == 1 || UGS_XD_HOSTMODE == 4), keep_ev = c && (UGS_XD_HOSTMODE == 1 || UGS_XD_HOSTMODE == 3);
    hipError_t e;
    if (keep_st) {
      if (!c->st && (e = hipStreamCreate(&c->st)) != hipSuccess) return e;
      st = c->st;
    } else { if ((e = hipStreamCreate(&st)) != hipSuccess) return e; own_st = true; }
    if (keep_ev) {
      if (!c->e0 && (e = hipEventCreate(&c->e0)) != hipSuccess) return e;
      if (!c->e1 && (e = hipEventCreate(&c->e1)) != hipSuccess) return e;
      e0 = c->e0; e1 = c->e1;
    } else {
      own_ev = true;
      if ((e = hipEventCreate(&e0)) != hipSuccess) return e;
      if ((e = hipEventCreate(&e1)) != hipSuccess) return e;
    }
    return hipSuccess;
  }
  template <class T> hipError_t alloc(T **p, size_t n)
  {
    void *q = nullptr;
    hipError_t e = ugs_malloc(&q, (n ? n : 1) * sizeof(T));
    if (e == hipSuccess) { ptrs.push_back(q); *p = (T *)q; }
    return e;
  }
};

}  // namespace

// letter class (a-z -> 0..25, '*' -> 26, anything else -> 27 scoring 0) and 2 x score per class pair
void ugs_xdrop_tables(int is_nucleo, float m2, float mm2, int8_t sub2[1024], uint8_t cls[256])
{
  memset(sub2, 0, 1024);
  for (int c = 0; c < 256; ++c) cls[c] = (c >= 'A' && c <= 'Z') ? (uint8_t)(c - 'A') : (c >= 'a' && c <= 'z') ? (uint8_t)(c - 'a') : (c == '*' ? 26 : 27);
  if (is_nucleo) {                                              // setnucmx.cpp:11-99: ACGTU by identity (T == U), others 0
    const char *alpha = "ACGTU";
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) {
      const bool same = (alpha[i] == alpha[j]) || ((alpha[i] == 'T' || alpha[i] == 'U') && (alpha[j] == 'T' || alpha[j] == 'U'));
      sub2[((alpha[i] - 'A') << 5) | (alpha[j] - 'A')] = (int8_t)(same ? m2 : mm2);
    }
  } else {                                                      // blosum62.cpp:22-84 incl. '*'
    for (int i = 0; i < 23; ++i) for (int j = 0; j < 23; ++j)
      sub2[((UGS_B62_ORDER[i] - 'A') << 5) | (UGS_B62_ORDER[j] - 'A')] = (int8_t)(2 * UGS_B62[i][j]);
    for (int i = 0; i < 23; ++i) { sub2[((UGS_B62_ORDER[i] - 'A') << 5) | 26] = -8; sub2[(26 << 5) | (UGS_B62_ORDER[i] - 'A')] = -8; }
    sub2[(26 << 5) | 26] = 2;
  }
}

#define XCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ugs_set_error("%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return UGS_E_HIP; } } while (0)

extern "C" void ugs_xdrop_params_init(ugs_xdrop_params *p, int is_nucleo)
{
  memset(p, 0, sizeof *p);
  p->is_nucleo = is_nucleo;
  p->match = 1.0f; p->mismatch = -2.0f;
  p->local_open = -10.0f;     // alnparams.cpp:362-369 with the o_defaults.inc:3-4 defaults (both alphabets)
  p->local_ext = -1.0f;
  p->xdrop = 32.0f;           // o_defaults.inc:20
}

extern "C" int ugs_xdrop_last_stats(float *ms_kernel, uint64_t *dp_cells)
{
  if (ms_kernel) *ms_kernel = g_last_ms;
  if (dp_cells) *dp_cells = g_last_cells;
  return UGS_OK;
}

extern "C" int ugs_xdrop_batch(int device, const ugs_xdrop_params *p,
                               const char *a_seqs, const uint64_t *a_offs, uint32_t na,
                               const char *b_seqs, const uint64_t *b_offs, uint32_t nb,
                               const ugs_xdrop_job *jobs, uint32_t njobs,
                               ugs_xdrop_hsp *hsps, uint32_t *path_pool, uint64_t path_cap, uint64_t *path_used)
{
  g_last_ms = 0.0f; g_last_cells = 0;
  if (path_used) *path_used = 0;
  if (!p || !a_offs || !b_offs || (njobs && (!jobs || !hsps)) || (na && !a_seqs && a_offs[na]) || (nb && !b_seqs && b_offs[nb])) {
    ugs_set_error("null argument"); return UGS_E_ARG;
  }
  const int ndev = ugs_device_count();
  if (device < 0 || device >= ndev) {
    ugs_set_error("device %d not available (%d devices); there is no CPU fallback", device, ndev); return UGS_E_NODEVICE;
  }
  if (njobs == 0) return UGS_OK;
  // scores as exact half-units
  const float o2 = p->local_open * 2.0f, e2 = p->local_ext * 2.0f, m2 = p->match * 2.0f, mm2 = p->mismatch * 2.0f;
  if (o2 != floorf(o2) || e2 != floorf(e2) || !(o2 < 0) || !(e2 < 0) || o2 < -2000 || e2 < -2000) {
    ugs_set_error("local_open/local_ext must be negative multiples of 0.5"); return UGS_E_ENVELOPE;
  }
  if (p->is_nucleo && (m2 != floorf(m2) || mm2 != floorf(mm2) || fabsf(m2) > 120 || fabsf(mm2) > 120)) {
    ugs_set_error("match/mismatch must be multiples of 0.5 within +-60 for the integer DP"); return UGS_E_ENVELOPE;
  }
  // validate the jobs the way the reference's asserts would (xdropalignmem.cpp:51-62, xdropfwdmem.cpp:354-355)
  uint32_t maxLB = 2, maxLA = 2, maxsum = 4;
  uint64_t arena_first = 0, arena_worst = 0;
  for (uint32_t k = 0; k < njobs; ++k) {
    const ugs_xdrop_job &j = jobs[k];
    if (j.a >= na || j.b >= nb) { ugs_set_error("job %u: sequence index out of range", k); return UGS_E_ARG; }
    const uint64_t la = a_offs[j.a + 1] - a_offs[j.a], lb = b_offs[j.b + 1] - b_offs[j.b];
    if (la == 0 || lb == 0 || la > 0x7fffffffull || lb > 0x7fffffffull) { ugs_set_error("job %u: empty or oversized sequence", k); return UGS_E_ARG; }
    uint32_t sideA, sideB;
    if (j.mode == UGS_XDROP_ALIGN) {
      if (j.anc_len > 1 && !(j.anc_loi < la && j.anc_loj < lb && (uint64_t)j.anc_loi + j.anc_len <= la && (uint64_t)j.anc_loj + j.anc_len <= lb)) {
        ugs_set_error("job %u: anchor outside the sequences", k); return UGS_E_ARG;
      }
      sideA = (uint32_t)std::min<uint64_t>(la, XD_MAXL + 1); sideB = (uint32_t)std::min<uint64_t>(lb, XD_MAXL + 1);
    } else if (j.mode == UGS_XDROP_FWD || j.mode == UGS_XDROP_BWD) {
      if (la > XD_MAXL + 1 || lb > XD_MAXL + 1) { ugs_set_error("job %u: FWD/BWD need both lengths <= %u", k, XD_MAXL + 1); return UGS_E_ENVELOPE; }
      sideA = (uint32_t)la; sideB = (uint32_t)lb;
    } else { ugs_set_error("job %u: bad mode", k); return UGS_E_ARG; }
    maxLA = std::max(maxLA, sideA); maxLB = std::max(maxLB, sideB);
    maxsum = std::max<uint64_t>(maxsum, la + lb + 4) > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)std::max<uint64_t>(maxsum, la + lb + 4);
    arena_first += std::min<uint64_t>(la + lb + 2, 96);
    arena_worst += la + lb + 2;
  }

  XCHK(hipSetDevice(device));
  DevBufs d;
  XCHK(d.open(device));
  hipDeviceProp_t prop;
  XCHK(hipGetDeviceProperties(&prop, device));

  int8_t sub2[1024]; uint8_t cls[256];
  ugs_xdrop_tables(p->is_nucleo, m2, mm2, sub2, cls);

  XdView v;
  memset(&v, 0, sizeof v);
  uint8_t *dA, *dB; uint64_t *dAo, *dBo; ugs_xdrop_job *dJ; ugs_xdrop_hsp *dH; uint32_t *dS;
  int8_t *dSub; uint8_t *dCls; unsigned int *dNext; unsigned long long *dCtr;
  const uint64_t nA = a_offs[na], nB = b_offs[nb];
  XCHK(d.alloc(&dA, nA + 64)); XCHK(d.alloc(&dB, nB + 64));
  XCHK(d.alloc(&dAo, (size_t)na + 1)); XCHK(d.alloc(&dBo, (size_t)nb + 1));
  XCHK(d.alloc(&dJ, njobs)); XCHK(d.alloc(&dH, njobs)); XCHK(d.alloc(&dS, njobs));
  XCHK(d.alloc(&dSub, 1024)); XCHK(d.alloc(&dCls, 256)); XCHK(d.alloc(&dNext, 4)); XCHK(d.alloc(&dCtr, 4));
  if (nA) XCHK(hipMemcpyAsync(dA, a_seqs, nA, hipMemcpyHostToDevice, d.st));
  if (nB) XCHK(hipMemcpyAsync(dB, b_seqs, nB, hipMemcpyHostToDevice, d.st));
  XCHK(hipMemcpyAsync(dAo, a_offs, ((size_t)na + 1) * 8, hipMemcpyHostToDevice, d.st));
  XCHK(hipMemcpyAsync(dBo, b_offs, ((size_t)nb + 1) * 8, hipMemcpyHostToDevice, d.st));
  XCHK(hipMemcpyAsync(dJ, jobs, (size_t)njobs * sizeof(ugs_xdrop_job), hipMemcpyHostToDevice, d.st));
  XCHK(hipMemcpyAsync(dSub, sub2, 1024, hipMemcpyHostToDevice, d.st));
  XCHK(hipMemcpyAsync(dCls, cls, 256, hipMemcpyHostToDevice, d.st));
  XCHK(hipMemsetAsync(dS, 0, (size_t)njobs * 4, d.st));
  XCHK(hipMemsetAsync(dCtr, 0, 32, d.st));
  v.a_seqs = dA; v.a_offs = dAo; v.b_seqs = dB; v.b_offs = dBo; v.jobs = dJ; v.hsps = dH; v.status = dS;
  v.sub2 = dSub; v.cls = dCls; v.next_job = dNext; v.arena_used = dCtr; v.cells = dCtr + 1;
  v.open2 = (int)o2; v.ext2 = (int)e2; v.X = p->xdrop; v.abs_open = -p->local_open; v.abs_ext = -p->local_ext;
  v.W = maxLB + 4;
  v.rows_cap = maxLA + 2;
  v.runbuf_cap = maxsum;

  const size_t wave_lds = (((size_t)v.W * 13) + 15) & ~(size_t)15;
  const unsigned long long tb_worst = (unsigned long long)(maxLA + 1) * (maxLB + 3) + 128;
  const uint64_t tb_budget = 24ull << 30;                       // HBM the traceback scratch may take
  std::vector<uint32_t> status(njobs);
  float ms_total = 0.0f;
  // one launch over n jobs (dList == nullptr: all of them) with the given scratch sizes
  auto launch = [&](const uint32_t *dList, uint32_t n, int wpb, unsigned long long tb_cap, uint32_t *dArena,
                    unsigned long long arena_cap, unsigned long long arena_start) -> int {
    const size_t lds = 1280 + wave_lds * wpb;
    XCHK(hipFuncSetAttribute((const void *)k_xdrop, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    XCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_xdrop, wpb * 64, lds));
    if (per_cu < 1) { ugs_set_error("x-drop kernel does not fit: %zu bytes of LDS", lds); return UGS_E_ENVELOPE; }
    uint64_t waves = (uint64_t)per_cu * prop.multiProcessorCount * wpb;
    waves = std::min<uint64_t>(waves, std::max<uint64_t>(1, tb_budget / tb_cap));
    waves = std::min<uint64_t>(waves, n);
    const uint32_t grid = (uint32_t)((waves + wpb - 1) / wpb);
    const uint64_t nw = (uint64_t)grid * wpb;
    uint8_t *dTB; uint2 *dRI; uint32_t *dRB;
    XCHK(d.alloc(&dTB, (size_t)(nw * tb_cap))); XCHK(d.alloc(&dRI, (size_t)(nw * v.rows_cap)));
    XCHK(d.alloc(&dRB, (size_t)(nw * 2 * v.runbuf_cap)));
    v.tb = dTB; v.tb_cap = tb_cap; v.rowinfo = dRI; v.runbuf = dRB; v.arena = dArena; v.arena_cap = arena_cap;
    v.job_list = dList; v.njobs = n;
    XCHK(hipMemsetAsync(dNext, 0, 4, d.st));
    XCHK(hipMemcpyAsync(dCtr, &arena_start, 8, hipMemcpyHostToDevice, d.st));
    XCHK(hipEventRecord(d.e0, d.st));
    hipLaunchKernelGGL(k_xdrop, dim3(grid), dim3(wpb * 64), lds, d.st, v);
    XCHK(hipGetLastError());
    XCHK(hipEventRecord(d.e1, d.st));
    XCHK(hipMemcpyAsync(status.data(), dS, (size_t)njobs * 4, hipMemcpyDeviceToHost, d.st));
    XCHK(hipStreamSynchronize(d.st));
    float ms = 0.0f;
    XCHK(hipEventElapsedTime(&ms, d.e0, d.e1));
    ms_total += ms;
    return UGS_OK;
  };
  // pass 0: scratch sized for typical jobs; jobs that overflow it are redone with worst-case scratch
  uint32_t *dArena0;
  XCHK(d.alloc(&dArena0, (size_t)arena_first));
  int rc = launch(nullptr, njobs, wave_lds <= 16384 ? 4 : 1, std::min<unsigned long long>(tb_worst, 1ull << 20), dArena0, arena_first, 0);
  if (rc != UGS_OK) return rc;
  std::vector<uint32_t> todo;
  uint64_t redo_worst = 0;
  for (uint32_t k = 0; k < njobs; ++k) {
    if (status[k] == ST_BAD) { ugs_set_error("job %u: internal run buffer overflow", k); return UGS_E_HIP; }
    if (status[k] != ST_OK) {
      todo.push_back(k);
      redo_worst += (a_offs[jobs[k].a + 1] - a_offs[jobs[k].a]) + (b_offs[jobs[k].b + 1] - b_offs[jobs[k].b]) + 2;
    }
  }
  if (!todo.empty()) {
    uint32_t *dComb, *dList;
    XCHK(d.alloc(&dComb, (size_t)(arena_first + redo_worst)));
    XCHK(hipMemcpyAsync(dComb, dArena0, (size_t)arena_first * 4, hipMemcpyDeviceToDevice, d.st));
    XCHK(d.alloc(&dList, todo.size()));
    XCHK(hipMemcpyAsync(dList, todo.data(), todo.size() * 4, hipMemcpyHostToDevice, d.st));
    rc = launch(dList, (uint32_t)todo.size(), 1, tb_worst, dComb, arena_first + redo_worst, arena_first);
    if (rc != UGS_OK) return rc;
    for (uint32_t k : todo) if (status[k] != ST_OK) { ugs_set_error("job %u: x-drop scratch exhausted (status %u)", k, status[k]); return UGS_E_NOMEM; }
  }
  g_last_ms = ms_total;
  unsigned long long cells = 0;
  XCHK(hipMemcpy(&cells, v.cells, 8, hipMemcpyDeviceToHost));
  g_last_cells = cells;

  // job-ordered pool
  uint64_t *dLen, *dOff;
  XCHK(d.alloc(&dLen, (size_t)njobs + 1)); XCHK(d.alloc(&dOff, (size_t)njobs + 1));
  XCHK(hipMemsetAsync(dLen, 0, ((size_t)njobs + 1) * 8, d.st));
  hipLaunchKernelGGL(k_xd_lens, dim3((njobs + 255) / 256), dim3(256), 0, d.st, dH, njobs, dLen);
  size_t tmp_bytes = 0;
  XCHK(rocprim::exclusive_scan(nullptr, tmp_bytes, dLen, dOff, (uint64_t)0, (size_t)njobs + 1, rocprim::plus<uint64_t>(), d.st));
  void *dTmp; XCHK(d.alloc((unsigned char **)&dTmp, tmp_bytes));
  XCHK(rocprim::exclusive_scan(dTmp, tmp_bytes, dLen, dOff, (uint64_t)0, (size_t)njobs + 1, rocprim::plus<uint64_t>(), d.st));
  uint64_t total = 0;
  XCHK(hipMemcpyAsync(&total, dOff + njobs, 8, hipMemcpyDeviceToHost, d.st));
  XCHK(hipStreamSynchronize(d.st));
  if (path_used) *path_used = total;
  if (total > path_cap || (total && !path_pool)) { ugs_set_error("path pool too small: need %llu runs", (unsigned long long)total); return UGS_E_CAPACITY; }
  uint32_t *dPool;
  XCHK(d.alloc(&dPool, (size_t)total));
  hipLaunchKernelGGL(k_xd_compact, dim3((uint32_t)(((uint64_t)njobs * 64 + 255) / 256)), dim3(256), 0, d.st, v.arena, dH, dOff, njobs, dPool);
  XCHK(hipGetLastError());
  XCHK(hipMemcpyAsync(hsps, dH, (size_t)njobs * sizeof(ugs_xdrop_hsp), hipMemcpyDeviceToHost, d.st));
  if (total) XCHK(hipMemcpyAsync(path_pool, dPool, (size_t)total * 4, hipMemcpyDeviceToHost, d.st));
  XCHK(hipStreamSynchronize(d.st));
  return UGS_OK;
}
