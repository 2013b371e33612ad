// ugs_index.hip - DB soft-masking and UDB word-index construction on the GPU (gfx950).
//
// Replaces (reference, /root/reference/src):
//   FastMaskSeq                      fastmask.cpp:88-158     (via MaskDB makeudb.cpp:11-25)
//   UDBData::FromSeqDB               udbbuild.cpp:303-398    (+ AddSeqNoncoded :256-284,
//   UDBParams::SeqToWordNoPattern    udbparams.cpp:540-555     SetTargetUniqueWords :680-711)
//
// Layout produced (all resident in HBM): CSR rows `postings[row_off[w] .. row_off[w+1])` =
// ascending target indexes containing word w (each target once per distinct valid word), plus
// a partition table part[w][p] = offset inside row w of the first target >= p * gsize, which
// lets the ranking kernel cut every row into LDS-sized target ranges without searching.
//
// Method: one 64-bit key (word << 32 | target) per sequence position, device radix sort,
// adjacent-unique; sort order == (word asc, target asc) == the reference's sequential insert order.
#include <cstring>
#include <algorithm>
#include "ugs_dev.h"
#ifndef RCCHK
#define RCCHK(x) do { int rc_ = (x); if (rc_ != UGS_OK) return rc_; } while (0)
#endif
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/device/device_scan.hpp>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ugs_set_error("%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return UGS_E_HIP; } } while (0)

// One thread per sequence: the reference's run detector is inherently sequential per sequence
// (a few hundred letters), and the whole DB is masked once at load.
__global__ void k_mask(uint8_t *seqs, const uint64_t *offs, uint32_t nseq, int dbmask)
{
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nseq) return;
  uint8_t *S = seqs + offs[t];
  const uint32_t L = (uint32_t)(offs[t + 1] - offs[t]);
  for (uint32_t i = 0; i < L; ++i) { uint8_t c = S[i]; if (c >= 'a' && c <= 'z') S[i] = c - 32; }
  if (!dbmask || L < 2) return;
  const uint8_t hard = (uint8_t)(dbmask >> 8);        // -hardmask: 'N' / 'X' instead of lower case (fastmask.cpp:98,117-122,145-150)
  // homopolymer runs: fastmask.cpp:106-130 (k1=5, j1=2); unsigned wrap of i-Start is intended
  {
    uint8_t last = '?';
    uint32_t start = 0xffffffffu;
    for (uint32_t i = 0; i < L; ++i) {
      uint8_t c = S[i]; if (c >= 'a' && c <= 'z') c -= 32;
      if (c != last || i + 1 == L) {
        uint32_t n1 = i - start;
        if (n1 >= 5) for (uint32_t j = start + 2; j < i; ++j) { uint8_t x = S[j]; if (hard) S[j] = hard; else if (x >= 'A' && x <= 'Z') S[j] = x + 32; }
        start = i;
      }
      last = c;
    }
  }
  // same-phase dinucleotide repeats: fastmask.cpp:132-157 (k2=5, j2=1), both phases, no flush at end
  for (uint32_t sp = 0; sp <= 1; ++sp) {
    uint32_t lastpair = 0xffffffffu, start = 0xffffffffu;
    for (uint32_t i = sp; i < L - 1; i += 2) {
      uint8_t c1 = S[i], c2 = S[i + 1];
      if (c1 >= 'a' && c1 <= 'z') c1 -= 32;
      if (c2 >= 'a' && c2 <= 'z') c2 -= 32;
      uint32_t pair = ((uint32_t)c1 << 8) + c2;
      if (pair != lastpair) {
        uint32_t n2 = i - start;
        // (the hard-mask branch starts one letter earlier: Start + j2 against Start + 2*j2, fastmask.cpp:146-151)
        if (n2 >= 5) for (uint32_t j = start + (hard ? 1 : 2); j < i; ++j) { uint8_t x = S[j]; if (hard) S[j] = hard; else if (x >= 'A' && x <= 'Z') S[j] = x + 32; }
        start = i;
      }
      lastpair = pair;
    }
  }
}

int ugs_launch_mask(uint8_t *d_seqs, const uint64_t *d_offs, uint32_t nseq, int dbmask, hipStream_t st)
{
  if (dbmask == 2) return UGS_OK;      // letters of a .udb: already masked, used as stored (LoadUDB loaddb.cpp:100-125)
  if (nseq == 0) return UGS_OK;
  hipLaunchKernelGGL(k_mask, dim3((nseq + 255) / 256), dim3(256), 0, st, d_seqs, d_offs, nseq, dbmask);
  HIPCHK(hipGetLastError());
  return UGS_OK;
}

// The stored (masked) nt letters once more, 2 bits per letter + a plane of "not A/C/G/T/U" bits, 16 letters per word by global letter
// index (BASELINE north_star: letters packed 2-bit in HBM).  k_align fetches a target's letters for the seed search and the ungapped
// extension from these arrays - a quarter of the bytes and no per-letter table look-ups per pair; the byte array stays what the
// reference's SeqDB is (case, IUPAC letters: identities, masking, output) and is read for the pairs that reach the DP.
// Code of a letter = what k_align's s_sc table gives it: hsp_letter (alpha2.cpp order A,C,G,T/U = 0..3) for a letter that can be part
// of a word whatever its case, "other" for the rest.  One thread per word; words [word_lo, word_hi).
__global__ void k_pack_letters(const UgsTables *tab, const uint8_t *seqs, uint64_t word_lo, uint64_t word_hi, uint2 *pk)
{
  __shared__ uint8_t code[256];
  for (int k = threadIdx.x; k < 256; k += blockDim.x) {
    const uint8_t cl = tab->cls[k] & 31;
    code[k] = (cl < 26 && tab->udb_letter['A' + cl] != 0xff) ? tab->hsp_letter['A' + cl] : 4;
  }
  __syncthreads();
  const uint64_t w = word_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= word_hi) return;
  const uint4 d4 = *(const uint4 *)(seqs + 16 * w);
  const uint32_t d[4] = {d4.x, d4.y, d4.z, d4.w};
  uint32_t v = 0, iv = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const uint32_t sc = code[(d[q] >> (8 * b)) & 0xffu];
      v |= (sc & 3u) << (8 * q + 2 * b);
      iv |= (sc >> 2) << (8 * q + 2 * b);
    }
  pk[w] = make_uint2(v, iv);
}

int ugs_launch_pack(const UgsTables *d_tab, const uint8_t *d_seqs, uint64_t word_lo, uint64_t word_hi, uint2 *d_pk, hipStream_t st)
{
  if (word_hi <= word_lo) return UGS_OK;
  const uint64_t n = word_hi - word_lo;
  hipLaunchKernelGGL(k_pack_letters, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, d_tab, d_seqs, word_lo, word_hi, d_pk);
  HIPCHK(hipGetLastError());
  return UGS_OK;
}

// One wave per sequence; lane = position.  key = word<<32 | target, or slots<<32 (sorts last)
// when the position has no valid word.
__global__ void k_word_keys(const UgsTables *tab, const uint8_t *seqs, const uint64_t *offs, uint32_t nseq,
                            int W, int alpha, uint32_t slots, uint64_t *keys)
{
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  if (wave >= nseq) return;
  const uint64_t o = offs[wave];
  const uint32_t L = (uint32_t)(offs[wave + 1] - o);
  const uint8_t *S = seqs + o;
  const uint64_t bad = (uint64_t)slots << 32;
  for (uint32_t pos = lane; pos < L; pos += 64) {
    uint64_t key = bad;
    if (pos + W <= L) {
      uint32_t w = 0; bool ok = true;
      for (int k = 0; k < W; ++k) {
        uint32_t l = tab->udb_letter[S[pos + k]];
        ok = ok && (l != 0xff);
        w = w * alpha + l;
      }
      if (ok) key = ((uint64_t)w << 32) | wave;
    }
    keys[o + pos] = key;
  }
}

__global__ void k_row_off(const uint64_t *ukeys, uint64_t n, uint32_t slots, uint64_t *row_off)
{
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > slots) return;
  const uint64_t want = (uint64_t)s << 32;
  uint64_t lo = 0, hi = n;
  while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (ukeys[mid] < want) lo = mid + 1; else hi = mid; }
  row_off[s] = lo;
}

__global__ void k_postings(const uint64_t *ukeys, uint64_t n, uint32_t *postings)
{
  uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) postings[k] = (uint32_t)ukeys[k];
}

__global__ void k_max_row(const uint64_t *row_off, uint32_t slots, uint32_t *max_row)
{
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t v = 0;
  if (s < slots) v = (uint32_t)(row_off[s + 1] - row_off[s]);
  for (int o = 32; o > 0; o >>= 1) { uint32_t x = __shfl_down(v, o); v = x > v ? x : v; }
  if ((threadIdx.x & 63) == 0 && v) atomicMax(max_row, v);
}

// Temporaries of ugs_build_index: cluster_fast builds two small indexes per batch (the batch's own and the new centroids'), and
// eight hipMalloc / hipFree pairs per build cost more than the kernels.  Buffers up to 64 MB are kept per host thread and device
// and reused; larger ones (a whole database) are allocated and released as before.
namespace {
struct Scratch {
  void *p = nullptr; size_t cap = 0; int dev = -1;
  ~Scratch() { /* released with the process: the device may be gone when thread-locals are destroyed */ }
};
constexpr size_t SCRATCH_KEEP = 64ull << 20;
thread_local Scratch g_scratch[5];
int scratch_get(int i, size_t bytes, void **out, bool *cached)
{
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (bytes > SCRATCH_KEEP) { *cached = false; HIPCHK(ugs_malloc(out, bytes)); return UGS_OK; }
  Scratch &sc = g_scratch[i];
  if (sc.dev != dev || sc.cap < bytes) {
    if (sc.p) {                                               // the old buffer may belong to the device this thread used before
      if (sc.dev != dev) HIPCHK(hipSetDevice(sc.dev));
      const hipError_t e = ugs_free(sc.p);
      if (sc.dev != dev) HIPCHK(hipSetDevice(dev));
      HIPCHK(e);
    }
    sc.p = nullptr; sc.cap = 0; sc.dev = dev;
    const size_t want = bytes + bytes / 4 + 4096;
    HIPCHK(ugs_malloc(&sc.p, want));
    sc.cap = want;
  }
  *cached = true; *out = sc.p;
  return UGS_OK;
}
}  // namespace

int ugs_build_index(const UgsTables *d_tab, const uint8_t *d_seqs, const uint64_t *d_offs, uint32_t nseq,
                    uint64_t nletters, int word_len, int alpha, uint32_t slots, uint64_t **d_row_off_out,
                    uint32_t **d_postings_out, uint64_t *n_postings, uint32_t *max_row, hipStream_t st)
{
  uint64_t *row_off = nullptr; uint32_t *postings = nullptr;
  HIPCHK(ugs_malloc(&row_off, ((size_t)slots + 1) * sizeof(uint64_t)));
  *d_row_off_out = row_off; *d_postings_out = nullptr; *n_postings = 0; *max_row = 0;
  if (nseq == 0 || nletters == 0) {
    HIPCHK(hipMemsetAsync(row_off, 0, ((size_t)slots + 1) * sizeof(uint64_t), st));
    HIPCHK(ugs_malloc(&postings, 256 * sizeof(uint32_t)));
    *d_postings_out = postings;
    return UGS_OK;
  }
  uint64_t *keys = nullptr, *keys2 = nullptr; unsigned long long *d_count = nullptr; void *tmp = nullptr; uint32_t *d_max = nullptr;
  bool c_keys = false, c_keys2 = false, c_cnt = false, c_tmp = false;
  RCCHK(scratch_get(0, nletters * sizeof(uint64_t), (void **)&keys, &c_keys));
  RCCHK(scratch_get(1, nletters * sizeof(uint64_t), (void **)&keys2, &c_keys2));
  RCCHK(scratch_get(2, 64, (void **)&d_count, &c_cnt));           // the counter and, 16 bytes on, the row maximum
  d_max = (uint32_t *)((char *)d_count + 16);
  HIPCHK(hipMemsetAsync(d_max, 0, sizeof(uint32_t), st));
  {
    uint64_t threads = (uint64_t)nseq * 64;
    hipLaunchKernelGGL(k_word_keys, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, d_tab, d_seqs, d_offs,
                       nseq, word_len, alpha, slots, keys);
    HIPCHK(hipGetLastError());
  }
  unsigned wordbits = 1; while ((1ull << wordbits) <= slots) ++wordbits;   // values 0..slots
  size_t tmp_bytes = 0;
  size_t tmp2 = 0;
  HIPCHK(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys, keys2, (size_t)nletters, 0u, 32u + wordbits, st));
  HIPCHK(rocprim::unique(nullptr, tmp2, keys2, keys, d_count, (size_t)nletters, rocprim::equal_to<uint64_t>(), st));
  RCCHK(scratch_get(3, std::max<size_t>(std::max(tmp_bytes, tmp2), 16), &tmp, &c_tmp));     // one buffer serves both primitives (stream order)
  HIPCHK(rocprim::radix_sort_keys(tmp, tmp_bytes, keys, keys2, (size_t)nletters, 0u, 32u + wordbits, st));
  HIPCHK(rocprim::unique(tmp, tmp2, keys2, keys, d_count, (size_t)nletters, rocprim::equal_to<uint64_t>(), st));
  unsigned long long nuniq = 0;
  HIPCHK(hipMemcpyAsync(&nuniq, d_count, sizeof(nuniq), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  hipLaunchKernelGGL(k_row_off, dim3((slots + 1 + 255) / 256), dim3(256), 0, st, keys, (uint64_t)nuniq, slots, row_off);
  HIPCHK(hipGetLastError());
  uint64_t np_host = 0;
  HIPCHK(hipMemcpyAsync(&np_host, row_off + slots, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(ugs_malloc(&postings, (np_host + 256) * sizeof(uint32_t)));        // padded: rows are read in whole-wave units
  HIPCHK(hipMemsetAsync(postings + np_host, 0, 256 * sizeof(uint32_t), st));
  if (np_host) {
    hipLaunchKernelGGL(k_postings, dim3((unsigned)((np_host + 255) / 256)), dim3(256), 0, st, keys, np_host, postings);
    HIPCHK(hipGetLastError());
  }
  hipLaunchKernelGGL(k_max_row, dim3((slots + 255) / 256), dim3(256), 0, st, row_off, slots, d_max);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(max_row, d_max, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (!c_tmp) HIPCHK(ugs_free(tmp));
  if (!c_keys) HIPCHK(ugs_free(keys));
  if (!c_keys2) HIPCHK(ugs_free(keys2));
  if (!c_cnt) HIPCHK(ugs_free(d_count));
  *d_postings_out = postings;
  *n_postings = np_host;
  return UGS_OK;
}

// thread per (slot, p): lower_bound of p*gsize inside the row; p == np -> row size
__global__ void k_part(const uint64_t *row_off, const uint32_t *postings, uint32_t slots, uint32_t np,
                       uint32_t gsize, uint32_t *part)
{
  uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t total = (uint64_t)slots * (np + 1);
  if (id >= total) return;
  uint32_t s = (uint32_t)(id / (np + 1)), p = (uint32_t)(id % (np + 1));
  const uint64_t b = row_off[s];
  const uint32_t n = (uint32_t)(row_off[s + 1] - b);
  uint32_t lo = 0, hi = n;
  if (p == np) lo = n;
  else {
    const uint64_t want = (uint64_t)p * gsize;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if ((uint64_t)postings[b + mid] < want) lo = mid + 1; else hi = mid; }
  }
  part[id] = lo;
}

int ugs_build_part(const uint64_t *d_row_off, const uint32_t *d_postings, uint32_t slots, uint32_t np,
                   uint32_t gsize, uint32_t *d_part, hipStream_t st)
{
  uint64_t total = (uint64_t)slots * (np + 1);
  hipLaunchKernelGGL(k_part, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_row_off, d_postings, slots, np,
                     gsize, d_part);
  HIPCHK(hipGetLastError());
  return UGS_OK;
}


// ---- hit-table compaction (the device side of HitMgr bookkeeping, hitmgr.cpp:120-183): the
// alignment kernel leaves a fixed-size table hits[unit*max_accepts + k] + hit_n[unit]; the host only
// wants the hits, grouped by query in query order (strand 0 before strand 1).
__global__ void k_hits_per_query(const uint32_t *hit_n, uint32_t nq, uint32_t ns, uint32_t *qn)
{
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  uint32_t n = 0;
  for (uint32_t s = 0; s < ns; ++s) n += hit_n[q * ns + s];
  qn[q] = n;
}

__global__ void k_hits_compact(const uint32_t *hit_n, const ugs_hit *table, const uint32_t *qoff, uint32_t nq, uint32_t ns,
                               uint32_t ma, ugs_hit *out, uint32_t query_base)
{
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  uint32_t o = qoff[q];
  for (uint32_t s = 0; s < ns; ++s) {
    const uint32_t u = q * ns + s, n = hit_n[u];
    // (the hit's place in HitMgr's append order - candidate order, plus strand first - goes into the upper flag bits:
    //  GetTopHit breaks ties by it, hitmgr.cpp:398-415)
    for (uint32_t k = 0; k < n; ++k) { ugs_hit h = table[(uint64_t)u * ma + k]; h.query += query_base; h.flags |= (o - qoff[q]) << UGS_HIT_ORDER_SHIFT; out[o++] = h; }
  }
}

int ugs_compact_hits(const uint32_t *d_hit_n, const ugs_hit *d_table, uint32_t nq, uint32_t ns, uint32_t ma,
                     uint32_t *d_qn, uint32_t *d_qoff, ugs_hit *d_out, void *d_tmp, size_t tmp_bytes, uint32_t query_base,
                     hipStream_t st)
{
  if (nq == 0) return UGS_OK;
  hipLaunchKernelGGL(k_hits_per_query, dim3((nq + 255) / 256), dim3(256), 0, st, d_hit_n, nq, ns, d_qn);
  HIPCHK(hipGetLastError());
  size_t need = 0;
  HIPCHK(rocprim::exclusive_scan(nullptr, need, d_qn, d_qoff, 0u, (size_t)nq, rocprim::plus<uint32_t>(), st));
  if (need > tmp_bytes) { ugs_set_error("scan scratch too small"); return UGS_E_NOMEM; }
  HIPCHK(rocprim::exclusive_scan(d_tmp, need, d_qn, d_qoff, 0u, (size_t)nq, rocprim::plus<uint32_t>(), st));
  hipLaunchKernelGGL(k_hits_compact, dim3((nq + 255) / 256), dim3(256), 0, st, d_hit_n, d_table, d_qoff, nq, ns, ma, d_out, query_base);
  HIPCHK(hipGetLastError());
  return UGS_OK;
}

// ---- the same for hit tables with overflow blocks (deep walks, UgsBatchView::xpool): hit k >= ma of unit u is entry (k - ma) % UGS_XBLOCK
// of block (k - ma) / UGS_XBLOCK of the unit's chain (UgsWalkState::xhead, xnext[])
__global__ void k_hits_copy_x(const uint32_t *hit_n, const ugs_hit *table, const uint32_t *qoff, uint32_t nq, uint32_t ns, uint32_t ma, ugs_hit *out,
                              uint32_t query_base, UgsXHits x)
{
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  uint32_t o = qoff[q];
  for (uint32_t s = 0; s < ns; ++s) {
    const uint32_t u = q * ns + s, n = hit_n[u];
    uint32_t blk = n > ma ? x.state[u].xhead : 0xffffffffu;
    for (uint32_t k = 0; k < n; ++k) {
      ugs_hit h;
      if (k < ma) h = table[(uint64_t)u * ma + k];
      else {
        const uint32_t e = (k - ma) % UGS_XBLOCK;
        if (e == 0 && k != ma) blk = x.next[blk];
        h = x.pool[(uint64_t)blk * UGS_XBLOCK + e];
      }
      h.query += query_base; h.flags |= (o - qoff[q]) << UGS_HIT_ORDER_SHIFT; out[o++] = h;
    }
  }
}

int ugs_count_hits(const uint32_t *d_hit_n, uint32_t nq, uint32_t ns, uint32_t *d_qn, uint32_t *d_qoff, void *d_tmp, size_t tmp_bytes, hipStream_t st)
{
  if (nq == 0) return UGS_OK;
  hipLaunchKernelGGL(k_hits_per_query, dim3((nq + 255) / 256), dim3(256), 0, st, d_hit_n, nq, ns, d_qn);
  HIPCHK(hipGetLastError());
  size_t need = 0;
  HIPCHK(rocprim::exclusive_scan(nullptr, need, d_qn, d_qoff, 0u, (size_t)nq, rocprim::plus<uint32_t>(), st));
  if (need > tmp_bytes) { ugs_set_error("scan scratch too small"); return UGS_E_NOMEM; }
  HIPCHK(rocprim::exclusive_scan(d_tmp, need, d_qn, d_qoff, 0u, (size_t)nq, rocprim::plus<uint32_t>(), st));
  return UGS_OK;
}

int ugs_copy_hits(const uint32_t *d_hit_n, const ugs_hit *d_table, uint32_t nq, uint32_t ns, uint32_t ma, const uint32_t *d_qoff, ugs_hit *d_out,
                  uint32_t query_base, const UgsXHits *x, hipStream_t st)
{
  if (nq == 0) return UGS_OK;
  if (x) hipLaunchKernelGGL(k_hits_copy_x, dim3((nq + 255) / 256), dim3(256), 0, st, d_hit_n, d_table, d_qoff, nq, ns, ma, d_out, query_base, *x);
  else hipLaunchKernelGGL(k_hits_compact, dim3((nq + 255) / 256), dim3(256), 0, st, d_hit_n, d_table, d_qoff, nq, ns, ma, d_out, query_base);
  HIPCHK(hipGetLastError());
  return UGS_OK;
}

size_t ugs_compact_tmp_bytes(uint32_t nq)
{
  size_t need = 0;
  uint32_t *p = nullptr;
  if (rocprim::exclusive_scan(nullptr, need, p, p, 0u, (size_t)(nq ? nq : 1), rocprim::plus<uint32_t>(), (hipStream_t)0) != hipSuccess) return 1 << 20;
  return need + 256;
}
