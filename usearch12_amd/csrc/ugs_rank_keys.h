// ugs_rank_keys.h - the ranking kernels' sortable candidate keys and the small path's pair filters, shared by ugs_rank.hip (the
// kernels that keep the K best candidates of a unit) and ugs_deep.hip (the complete sorted list of a unit whose walk needs more).  Internal.
#pragma once
#include "ugs_dev.h"

// key = (CMAXV - count) << POS_BITS | first-touch position: ascending keys = CountSort*Desc order (count descending, stable in the
// order of the scan: countsort.cpp:6-191).  Position: small path = the target (SetTop walks the targets in ascending order,
// udbusortedsearcher.cpp:205-267), Big path = first row << 32 | target (the first-touch list, udbusortedsearcherbig.cpp:82-100)
#define POS_BITS 44
#define POS_MASK ((1ull << POS_BITS) - 1)
#define CMAXV 4095u
__device__ __forceinline__ uint64_t make_key(uint32_t c, uint64_t pos) { return ((uint64_t)(CMAXV - c) << POS_BITS) | pos; }
__device__ __forceinline__ uint32_t key_count(uint64_t k) { return CMAXV - (uint32_t)(k >> POS_BITS); }
__device__ __forceinline__ uint32_t key_target(uint64_t k) { return (uint32_t)k; }

// Pair filters on the small ranking path (Accepter::RejectPair accepter.cpp:140-197): there a refused pair leaves no
// trace in the candidate walk (udbusortedsearcher.cpp:145-147 ignores SetTarget's result, searcher.cpp:63-67 returns
// before the terminator), so refused targets are simply not candidates: they are dropped where candidates are chosen.
// (-selfid needs the letters and stays in k_align.)
struct PairQ {
  uint32_t mask, ql, qkey, qsize;
  float min_sizeratio, minqt, maxqt, minsl, maxsl;
  const uint64_t *offs; const uint32_t *t_key, *t_size;
};
__device__ __forceinline__ bool pair_reject(const PairQ &q, uint32_t t)
{
  const uint32_t m = q.mask;
  if (m & (UGS_P_SELF | UGS_P_NOTSELF)) {
    const bool same = q.qkey == q.t_key[t];
    if (((m & UGS_P_SELF) && same) || ((m & UGS_P_NOTSELF) && !same)) return true;
  }
  if ((m & UGS_P_MIN_SIZERATIO) && (double)q.t_size[t] / (double)q.qsize < (double)q.min_sizeratio) return true;
  if (m & (UGS_P_MINQT | UGS_P_MAXQT | UGS_P_MINSL | UGS_P_MAXSL)) {
    const uint32_t tl = (uint32_t)(q.offs[t + 1] - q.offs[t]), ql = q.ql;
    const double qt = (double)ql / (double)tl, sl = (double)(ql < tl ? ql : tl) / (double)(ql > tl ? ql : tl);
    if (((m & UGS_P_MINQT) && qt < (double)q.minqt) || ((m & UGS_P_MAXQT) && qt > (double)q.maxqt) ||
        ((m & UGS_P_MINSL) && sl < (double)q.minsl) || ((m & UGS_P_MAXSL) && sl > (double)q.maxsl)) return true;
  }
  return false;
}

