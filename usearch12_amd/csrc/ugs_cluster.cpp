// ugs_cluster.cpp - cluster_fast (UCLUST greedy clustering) on the device search path: SURVEY.md 8f-3, BASELINE config C3.
//
// Reference (paths relative to /root/reference/src):
//   ClusterFast                     clusterfast.cpp:81-133   derep -> serial loop over the uniques in input order
//   DerepFull / DerepResult         derepfull.cpp:130-212, derepresult.cpp:403-480 (at -threads 1: uniques numbered by first member)
//   MakeClusterSearcher             makeclustersearcher.cpp:13-112 (terminator 1 accept / 8 rejects terminator.cpp:10-14, empty UDB)
//   ClusterSink::OnQueryDone        clustersink.cpp:306-359 (GetTopHit => member; none => new centroid)
//   UDBData::AddSIToDB_CopyData     udbbuild.cpp:286-291 (AddSeqNoncoded :256-284, AddWord/GrowRow :74-128)
//   UDBUsortedSearcher::SetQueryImpl udbusortedsearcher.cpp:39-58 (small -> Big latch, tested per strand search)
//   OutputSink::OutputUC / OutputUCNoHits outputuc.cpp:10-93, ClusterSink::WriteUC_CRecs / CentroidsToFASTA clustersink.cpp:262-289,477-493
//
// The reference loop is serial: query i is searched against the centroids founded by queries < i.  Here the uniques are
// processed in batches.  A batch is searched (k_rank / k_align, unchanged search semantics) against the index as it stood
// when the batch started ("frozen"), with two changes of detail: the candidate list is chosen without the CountSort
// MinValue cut-off and the walk records how far it went.  What the serial loop would have seen in addition are the
// centroids founded by EARLIER QUERIES OF THE SAME BATCH.  Word counts against those (k_inbatch) and their pairwise
// alignments (k_align over explicit pair lists) are computed on the device for every pair that can matter; a host pass in
// input order then replays, per query strand, the reference's candidate order exactly: prefix maxima of the merged scan ->
// NextValue / MinValue (countsort.cpp:13-24), -bump ratchet on the small path (udbusortedsearcher.cpp:230-267), count-desc /
// scan-order-asc merge of frozen and in-batch candidates, walk with the terminator.  Anything the replay cannot decide
// from what the device delivered ends the batch at that query (it becomes the first query of the next batch, where it has
// no in-batch candidates) - never a guess, never a CPU search.
#include "ugs_host.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include <chrono>
#include <thread>

static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int ugs_index_merge(const uint64_t *old_off, const uint32_t *old_post, const uint64_t *delta_off, const uint32_t *delta_post,
                    uint32_t slots, uint32_t base_target, uint64_t *new_off, uint32_t *new_post, uint64_t n_total,
                    uint32_t *d_max_row, hipStream_t st);
int ugs_launch_inbatch(const UgsBatchView &bv, const uint64_t *brow_off, const uint32_t *bpost, uint32_t ns_max, int small_path,
                       uint32_t max_rej, int num_cu, uint32_t *ent_n, const uint32_t *ent_off, uint2 *ent, hipStream_t st,
                       uint32_t slot_cap, const uint32_t *list, uint32_t n_list);
#define INBATCH_SLOTS 16u        // entries per unit the first in-batch launch delivers (a unit has about one; the few with more get a second launch)

#define POS_BITS 44
#define CMAXV 4095u
static inline uint64_t mk_key(uint32_t c, uint64_t pos) { return ((uint64_t)(CMAXV - c) << POS_BITS) | pos; }
static inline uint32_t key_cnt(uint64_t k) { return CMAXV - (uint32_t)(k >> POS_BITS); }
static inline uint64_t key_pos(uint64_t k) { return k & ((1ull << POS_BITS) - 1); }

// ------------------------------------------------------------------------------------------------- ugs_db_append
// UDBData::AddSIToDB_CopyData for n sequences at once: they get the indexes nseq .. nseq+n-1; letters are stored as given
// (the cluster database is never masked); the index rows of their distinct valid words grow at the end.
extern "C" int ugs_db_append(ugs_db *db, const char *seqs, const uint64_t *offs, uint32_t n)
{
  if (!db || !offs || (n && !seqs)) { ugs_set_error("null argument"); return UGS_E_ARG; }
  if (n == 0) return UGS_OK;
  if (db->p.dbmask != 2) { ugs_set_error("ugs_db_append needs dbmask = 2 (letters used as given; a masked database cannot grow)"); return UGS_E_ARG; }
  HIPCHK(hipSetDevice(db->device));
  hipStream_t st = db->stream;
  const uint32_t old_n = db->v.nseq;
  const uint64_t add = offs[n] - offs[0];
  uint32_t maxl = db->max_tlen;
  std::vector<uint64_t> abs_off(n + 1), rel_off(n + 1);
  for (uint32_t i = 0; i <= n; ++i) { rel_off[i] = offs[i] - offs[0]; abs_off[i] = db->nletters + rel_off[i]; }
  for (uint32_t i = 0; i < n; ++i) {
    const uint64_t L = rel_off[i + 1] - rel_off[i];
    if (L > 65535) { ugs_set_error("sequence longer than 65535 letters"); return UGS_E_ENVELOPE; }
    maxl = std::max<uint32_t>(maxl, (uint32_t)L);
  }
  // letters and offsets (amortised growth)
  if (db->nletters + add + 64 > db->seq_cap) {
    const uint64_t cap = (db->nletters + add) * 2 + 4096;
    uint8_t *p = nullptr;
    HIPCHK(ugs_malloc(&p, cap));
    if (db->nletters) HIPCHK(hipMemcpyAsync(p, db->d_seqs, db->nletters, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(ugs_free(db->d_seqs)); db->d_seqs = p; db->seq_cap = cap;
  }
  if ((uint64_t)old_n + n + 1 > db->off_cap) {
    const uint64_t cap = ((uint64_t)old_n + n) * 2 + 1024;
    uint64_t *p = nullptr;
    HIPCHK(ugs_malloc(&p, cap * 8));
    HIPCHK(hipMemcpyAsync(p, db->d_offs, ((size_t)old_n + 1) * 8, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(ugs_free(db->d_offs)); db->d_offs = p; db->off_cap = cap;
  }
  if (add) HIPCHK(ugs_h2d(db->d_seqs + db->nletters, seqs + offs[0], add, st));
  if (db->p.is_nucleo) {                          // the packed letters follow: every word the new letters touch is packed again from the bytes
    const uint64_t need = (db->nletters + add + 64) / 16 + 8;
    if (need > db->pack_cap) {
      const uint64_t cap = need * 2;
      uint2 *a = nullptr;
      HIPCHK(ugs_malloc(&a, cap * 8));
      HIPCHK(hipMemsetAsync(a, 0, cap * 8, st));
      if (db->d_pk) HIPCHK(hipMemcpyAsync(a, db->d_pk, db->pack_cap * 8, hipMemcpyDeviceToDevice, st));
      HIPCHK(hipStreamSynchronize(st));
      (void)ugs_free(db->d_pk);
      db->d_pk = a; db->pack_cap = cap;
    }
    RCCHK(ugs_launch_pack(db->d_tab, db->d_seqs, db->nletters / 16, (db->nletters + add + 15) / 16, db->d_pk, st));
  }
  HIPCHK(ugs_h2d(db->d_offs + old_n, abs_off.data(), ((size_t)n + 1) * 8, st));
  struct Tmp {                                    // released on every way out, error returns included
    uint64_t *rel = nullptr, *drow = nullptr; uint32_t *dpost = nullptr, *dmax = nullptr;
    ~Tmp() { (void)ugs_free(rel); (void)ugs_free(drow); (void)ugs_free(dpost); (void)ugs_free(dmax); }
  } t;
  HIPCHK(ugs_malloc(&t.rel, ((size_t)n + 1) * 8));
  HIPCHK(ugs_h2d(t.rel, rel_off.data(), ((size_t)n + 1) * 8, st));
  // index of the new sequences on their own (targets 0..n-1), then row-wise append
  uint64_t n_dpost = 0; uint32_t dmax = 0;
  int rc = ugs_build_index(db->d_tab, db->d_seqs + db->nletters, t.rel, n, add, db->p.word_len, db->v.alpha, db->v.slots, &t.drow,
                           &t.dpost, &n_dpost, &dmax, st);
  if (rc != UGS_OK) { (void)hipStreamSynchronize(st); return rc; }
  const uint64_t total = db->n_postings + n_dpost;
  // the merged index goes into the handle's spare arrays, which then change places with the live ones (a multi-GB
  // hipMalloc / hipFree per append would cost more than the merge)
  if (!db->d_row_off2) HIPCHK(ugs_malloc(&db->d_row_off2, ((size_t)db->v.slots + 1) * 8));
  if (!db->d_postings2 || total + 256 > db->post_cap2) {
    if (db->d_postings2) HIPCHK(ugs_free(db->d_postings2));
    db->d_postings2 = nullptr;
    db->post_cap2 = total + total / 2 + 4096 + 256;
    HIPCHK(ugs_malloc(&db->d_postings2, db->post_cap2 * 4));
  }
  HIPCHK(ugs_malloc(&t.dmax, 4));
  rc = ugs_index_merge(db->d_row_off, db->d_postings, t.drow, t.dpost, db->v.slots, old_n, db->d_row_off2, db->d_postings2, total, t.dmax, st);
  if (rc == UGS_OK && hipMemcpyAsync(&db->max_row, t.dmax, 4, hipMemcpyDeviceToHost, st) != hipSuccess) rc = UGS_E_HIP;
  if (hipStreamSynchronize(st) != hipSuccess) rc = UGS_E_HIP;
  if (rc != UGS_OK) { ugs_set_error("index append failed"); return rc; }
  std::swap(db->d_row_off, db->d_row_off2); std::swap(db->d_postings, db->d_postings2); std::swap(db->post_cap, db->post_cap2);
  db->n_postings = total; db->nletters += add; db->max_tlen = maxl;
  db->v.nseq = old_n + n;
  return ugs_db_replan(db);
}

// ------------------------------------------------------------------------------------------------- the clustering handle
struct ugs_cluster {
  ugs_params p;
  uint32_t nseq = 0, n_unique = 0, n_clusters = 0;
  std::vector<uint32_t> seq_unique, uniq_seed, uniq_size, uniq_cluster, uniq_nhits, centroid_uniq, cluster_size;
  std::vector<uint64_t> uniq_hit_off;
  std::vector<ugs_hit> hits;
  std::vector<uint32_t> pool;
  // of the input (borrowed during ugs_cluster_fast): every sequence's boundaries, and the centroids' letters in cluster order
  std::vector<char> seqs; std::vector<uint64_t> offs, cent_off;
  ugs_cluster_stats st;
};

extern "C" int ugs_params_set_cluster(ugs_params *p)
{
  if (!p) return UGS_E_ARG;
  p->max_accepts = 1; p->max_rejects = 8;        // Terminator(CMD_cluster_fast) terminator.cpp:10-14
  p->dbmask = 2;                                  // SeqDB::FromFastx keeps the letters as read; nothing masks them later
  return UGS_OK;
}

// SeqHash32 / SeqHashRC32 (seqhash.cpp:6-34) and SeqEq / SeqEqRC (:44-69, case-insensitive).  At -threads 1 DerepFull numbers
// the uniques by their first member in input order (derepfull.cpp:171-181 one bucket, derepresult.cpp:443-480), which is what
// any exact grouping in input order gives; the hash only has to group equal sequences.
namespace {
struct Comp { uint8_t c[256]; Comp() { for (int i = 0; i < 256; ++i) c[i] = (uint8_t)i; const char *f = "ABCDGHKMNRSTUVWXY", *t = "TVGHCDMKNYSAABWXR";
  for (int k = 0; f[k]; ++k) { c[(uint8_t)f[k]] = (uint8_t)t[k]; if (f[k] != 'U') c[(uint8_t)(f[k] | 0x20)] = (uint8_t)(t[k] | 0x20); } } };
const Comp g_comp;
inline uint8_t up(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }
uint32_t seq_hash(const uint8_t *s, uint32_t L, bool rc)
{
  uint32_t a = 63689, b = 378551, h = 0;
  for (uint32_t k = 0; k < L; ++k) { const uint8_t c = rc ? g_comp.c[s[L - k - 1]] : s[k]; h = h * a + up(c); a *= b; }
  return h;
}
bool seq_eq(const uint8_t *a, const uint8_t *b, uint32_t L) { for (uint32_t i = 0; i < L; ++i) if (up(a[i]) != up(b[i])) return false; return true; }
bool seq_eq_rc(const uint8_t *a, const uint8_t *b, uint32_t L) { for (uint32_t i = 0; i < L; ++i) if (up(a[i]) != up(g_comp.c[b[L - i - 1]])) return false; return true; }

uint32_t derep_full(const char *seqs, const uint64_t *offs, uint32_t nseq, bool revcomp, std::vector<uint32_t> &seq_unique, std::vector<uint32_t> &uniq_seed)
{
  seq_unique.assign(nseq, 0); uniq_seed.clear();
  // the hashes are independent of each other: computed by a few host threads; the grouping itself stays in input order
  std::vector<uint32_t> hv(nseq);
  {
    unsigned nt = std::min<unsigned>(16, std::max<unsigned>(1, std::thread::hardware_concurrency()));
    if (nseq < 100000) nt = 1;
    auto work = [&](unsigned t) {
      const uint32_t lo = (uint32_t)((uint64_t)nseq * t / nt), hi = (uint32_t)((uint64_t)nseq * (t + 1) / nt);
      for (uint32_t i = lo; i < hi; ++i) {
        const uint8_t *q = (const uint8_t *)seqs + offs[i];
        const uint32_t L = (uint32_t)(offs[i + 1] - offs[i]);
        uint32_t h = seq_hash(q, L, false);
        if (revcomp) h = std::min(h, seq_hash(q, L, true));
        hv[i] = h;
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
  }
  // Grouping.  The reference numbers the uniques by their first member in input order; WHICH reads are equal does not depend on
  // the order they are compared in, so the reads are split by hash among a few host threads, every thread groups its share in input
  // order with a table of its own (group = index of its first member), and the numbering follows from the first members in
  // ascending order.  (One thread, one table: 0.53 s of the 6.4 s of C3.)
  unsigned nt = std::min<unsigned>(16, std::max<unsigned>(1, std::thread::hardware_concurrency()));
  if (nseq < 100000) nt = 1;
  std::vector<uint32_t> first(nseq);
  auto group = [&](unsigned t) {
    uint64_t mine = 0;
    for (uint32_t i = 0; i < nseq; ++i) mine += (hv[i] % nt) == t;
    const uint64_t sl = mine * 2 + 7;
    std::vector<uint32_t> tb(sl, 0xffffffffu);
    for (uint32_t i = 0; i < nseq; ++i) {
      const uint32_t h = hv[i];
      if (h % nt != t) continue;
      const uint8_t *q = (const uint8_t *)seqs + offs[i];
      const uint32_t L = (uint32_t)(offs[i + 1] - offs[i]);
      uint64_t k = (h / nt) % sl;
      for (;;) {
        const uint32_t si = tb[k];
        if (si == 0xffffffffu) { tb[k] = i; first[i] = i; break; }
        if ((uint32_t)(offs[si + 1] - offs[si]) == L) {
          const uint8_t *us = (const uint8_t *)seqs + offs[si];
          if (seq_eq(q, us, L) || (revcomp && seq_eq_rc(q, us, L))) { first[i] = si; break; }
        }
        k = (k + 1) % sl;
      }
    }
  };
  {
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(group, t);
    group(0);
    for (auto &x : th) x.join();
  }
  for (uint32_t i = 0; i < nseq; ++i) if (first[i] == i) { seq_unique[i] = (uint32_t)uniq_seed.size(); uniq_seed.push_back(i); }
  for (uint32_t i = 0; i < nseq; ++i) if (first[i] != i) seq_unique[i] = seq_unique[first[i]];
  return (uint32_t)uniq_seed.size();
}

// sort.h:63-103 QuickSortOrderRecurse<unsigned, Desc = true> (ClusterSink::GetClusterSizeOrder clustersink.cpp:449-458)
void qs_order_desc_u(const uint32_t *V, int left, int right, uint32_t *Order)
{
  int i = left, j = right;
  const uint32_t pivot = V[Order[(left + right) / 2]];
  while (i <= j) {
    while (V[Order[i]] > pivot) i++;
    while (V[Order[j]] < pivot) j--;
    if (i <= j) { std::swap(Order[i], Order[j]); i++; j--; }
  }
  if (left < j) qs_order_desc_u(V, left, j, Order);
  if (i < right) qs_order_desc_u(V, i, right, Order);
}

struct DevBuf {            // a device array that only ever grows
  void *p = nullptr; size_t cap = 0;
  int need(size_t bytes) {
    if (bytes <= cap) return UGS_OK;
    if (p) (void)ugs_free(p);
    p = nullptr; cap = bytes + bytes / 2 + 4096;
    if (ugs_malloc(&p, cap) != hipSuccess) { cap = 0; ugs_set_error("ugs_malloc(%zu) failed", bytes); return UGS_E_HIP; }
    return UGS_OK;
  }
  ~DevBuf() { if (p) (void)ugs_free(p); }
};

enum { ST_UNKNOWN = 0, ST_MEMBER = 1, ST_CENTROID = 2, ST_MAYBE = 3 };
enum { R_NOHIT = 0, R_HIT_FROZEN = 1, R_HIT_BATCH = 2, R_NEED = 3, R_HARD = 4 };

struct UnitRes { int kind; uint32_t idx; };       // R_HIT_FROZEN: idx unused (hits[unit]); R_HIT_BATCH: idx = entry index

struct BatchHost {           // host copies of what the device delivered for one batch
  uint32_t nq = 0, ns = 1, K = 0;
  std::vector<uint64_t> cand_key; std::vector<uint32_t> cand, cand_n, walk_n, hit_n, cl_info; std::vector<uint64_t> cl_ev;
  std::vector<ugs_hit> fhits;               // [units] (max_accepts = 1)
  std::vector<uint32_t> ent_n, ent_off; std::vector<uint2> ent;
  std::vector<int8_t> pair_out;             // per entry: 0 unknown, 1 reject, 2 accept
  std::vector<uint32_t> pair_hit;           // per entry: index into phits
  std::vector<ugs_hit> phits;
  std::vector<uint32_t> pool;               // run pool shared by fhits and phits
};
}  // namespace

// The reference's candidate walk of one unit, replayed with the centroids of the same batch merged in.
//   status[j] of every earlier query of the batch; an in-batch entry takes part iff its sequence is (or, with
//   `potential`, may be) a centroid.  With potential = true nothing is decided: the caller only learns whether any entry
//   could take part (returns R_NEED) or none (falls through to the exact replay).
static UnitRes replay_unit(const BatchHost &H, const ugs_params &p, bool small_path, uint32_t unit, const uint8_t *status, bool potential)
{
  const uint32_t K = H.K;
  const uint32_t fn = H.cand_n[unit], wn = H.walk_n[unit], hn = H.hit_n[unit];
  const uint64_t *fkey = &H.cand_key[(uint64_t)unit * K];
  const uint32_t e0 = H.ent_off[unit], en = H.ent_n[unit];
  const uint32_t nev = H.cl_info[(uint64_t)unit * 4 + 2];
  struct V { uint64_t key; uint32_t e; };
  V vs[64]; std::vector<V> vbig; V *v = vs; uint32_t nv = 0;
  for (uint32_t e = e0; e < e0 + en; ++e) {
    const uint8_t s = status[H.ent[e].x];
    if (s == ST_MEMBER) continue;
    if (potential) return UnitRes{R_NEED, 0};
    if (s != ST_CENTROID) return UnitRes{R_HARD, 0};              // (an undecided earlier query: cannot happen in input order)
    const uint32_t j = H.ent[e].x, c = H.ent[e].y & 0xffffu, row = H.ent[e].y >> 16;
    const uint64_t pos = small_path ? (uint64_t)(0x80000000u | j) : (((uint64_t)row << 32) | (0x80000000u | j));
    if (nv == 64 && vbig.empty()) { vbig.assign(vs, vs + 64); }
    if (!vbig.empty()) { vbig.push_back(V{mk_key(c, pos), e}); v = vbig.data(); } else vs[nv] = V{mk_key(c, pos), e};
    ++nv;
  }
  if (!vbig.empty()) v = vbig.data();
  uint32_t min_value;
  if (nv == 0) min_value = H.cl_info[(uint64_t)unit * 4 + 1] / 2;   // the frozen scan's own NextValue / 2
  else {
    if (nev > UGS_CL_EV) return UnitRes{R_HARD, 0};
    // scan order: frozen prefix maxima (stored by descending position) and the in-batch entries, by position
    std::sort(v, v + nv, [](const V &a, const V &b) { return key_pos(a.key) < key_pos(b.key); });
    const uint64_t *ev = &H.cl_ev[(uint64_t)unit * UGS_CL_EV];
    uint32_t keep = nv;
    if (small_path && p.bump_pct != 0) {
      // SetTopBump (udbusortedsearcher.cpp:230-267): the frozen targets come first (smaller indexes); their prefix maxima
      // are the only elements that move MinU.  Then the in-batch centroids in index order.
      const double Bump = p.bump_pct / 100.0;
      uint32_t MinU = 1, MaxCount = 0;
      for (int e = (int)nev - 1; e >= 0; --e) {
        const uint32_t n = (uint32_t)(ev[e] >> POS_BITS);
        const uint32_t NewMin = (uint32_t)(n * Bump);
        if (NewMin > MinU && NewMin < MaxCount) MinU = NewMin;
        MaxCount = n;
      }
      keep = 0;
      for (uint32_t k = 0; k < nv; ++k) {
        const uint32_t n = key_cnt(v[k].key);
        if (n >= MinU) {
          if (n > MaxCount) { const uint32_t NewMin = (uint32_t)(n * Bump); if (NewMin > MinU && NewMin < MaxCount) MinU = NewMin; MaxCount = n; }
          v[keep++] = v[k];
        }
      }
    }
    nv = keep;
    // CountSort*Desc (countsort.cpp:13-24,114-126): NextValue = the running maximum just before it last rose
    uint32_t Max = 0, Next = 0;
    int e = (int)nev - 1; uint32_t k = 0;
    while (e >= 0 || k < nv) {
      uint32_t val;
      if (e >= 0 && (k >= nv || (ev[e] & ((1ull << POS_BITS) - 1)) < key_pos(v[k].key))) { val = (uint32_t)(ev[e] >> POS_BITS); --e; }
      else { val = key_cnt(v[k].key); ++k; }
      if (val > Max) { Next = Max; Max = val; }
    }
    min_value = Next / 2;
    std::sort(v, v + nv, [](const V &a, const V &b) { return a.key < b.key; });
  }
  // the walk: candidates by (count desc, scan position asc), counts below MinValue are not candidates
  uint32_t fi = 0, vi = 0, rej = 0;
  for (;;) {
    const bool fok = fi < fn, vok = vi < nv;
    if (!fok && !vok) return UnitRes{R_NOHIT, 0};
    bool take_f = fok && (!vok || fkey[fi] < v[vi].key);
    const uint64_t key = take_f ? fkey[fi] : v[vi].key;
    if (key_cnt(key) < min_value) return UnitRes{R_NOHIT, 0};       // everything behind has a count at most this one
    bool acc;
    if (take_f) {
      if (fi >= wn) return UnitRes{R_HARD, 0};                      // the device walk did not go this far (cannot happen, see DESIGN)
      acc = (fi == wn - 1) && hn != 0;
      if (acc) return UnitRes{R_HIT_FROZEN, 0};
      ++fi;
    } else {
      const int8_t o = H.pair_out[v[vi].e];
      if (o == 0) return UnitRes{R_NEED, v[vi].e};
      if (o == 2) return UnitRes{R_HIT_BATCH, v[vi].e};
      ++vi;
    }
    if (++rej == (uint32_t)p.max_rejects) return UnitRes{R_NOHIT, 0};
  }
}

extern "C" void ugs_cluster_destroy(ugs_cluster *c) { delete c; }

extern "C" int ugs_cluster_fast(const ugs_params *pp, const char *seqs, const uint64_t *offs, uint32_t nseq, int device, ugs_cluster **out)
{
  return ugs_cluster_fast_sorted(pp, seqs, offs, nseq, UGS_SORT_NONE, nullptr, 0, device, out);
}

// GetSizeFromLabel label.cpp:152-161
extern "C" uint32_t ugs_label_size(const char *label)
{
  const char *z = label ? strstr(label, ";size=") : nullptr;
  return z ? (uint32_t)atoi(z + 6) : 0xffffffffu;
}

extern "C" int ugs_cluster_fast_sorted(const ugs_params *pp, const char *seqs, const uint64_t *offs, uint32_t nseq, int sort_mode,
                                       const uint32_t *size_in, int sizein, int device, ugs_cluster **out)
{
  if (!pp || !offs || !out || (nseq && !seqs)) { ugs_set_error("null argument"); return UGS_E_ARG; }
  if (sort_mode != UGS_SORT_NONE && sort_mode != UGS_SORT_LENGTH && sort_mode != UGS_SORT_SIZE) { ugs_set_error("Invalid sort name"); return UGS_E_ARG; }   // clusterfast.cpp:65-66
  if (sizein) {                                                                                    // GetSizeFromLabel(Label, UINT_MAX) label.cpp:158-159
    if (!size_in) { ugs_set_error("-sizein needs the labels' size= values"); return UGS_E_ARG; }
    for (uint32_t i = 0; i < nseq; ++i) if (size_in[i] == 0xffffffffu) { ugs_set_error("Missing size= in the label of input sequence %u", i); return UGS_E_ARG; }
  }
  if (!pp->id_set) { ugs_set_error("Must specify -id"); return UGS_E_ARG; }                     // makeclustersearcher.cpp:30-31
  if (pp->max_accepts != 1 || pp->dbmask != 2 || pp->local || pp->pair_mask || pp->align_flags || (pp->filter_mask & UGS_F_ABSKEW)) {
    ugs_set_error("cluster_fast: use ugs_params_set_cluster (one accept per strand, letters as read); pair filters / -fulldp / -termid are not supported here");
    return UGS_E_ENVELOPE;
  }
  if (pp->max_rejects < 1 || pp->max_rejects > UGS_KMAX) {
    // (the greedy loop's walk records hold the UGS_KMAX candidates of a ranking pass: deeper walks exist for usearch_global only, ugs_deep.hip)
    ugs_set_error("cluster_fast: max_rejects must be 1 .. %d", UGS_KMAX); return UGS_E_ENVELOPE;
  }
  if (nseq == 0) { ugs_set_error("No sequences in input file"); return UGS_E_ARG; }                // clusterfast.cpp:91-92
  const ugs_params p = *pp;
  ugs_cluster *C = new ugs_cluster();
  C->p = p; C->nseq = nseq;
  memset(&C->st, 0, sizeof(C->st));
  // the input is borrowed for the duration of the call; the handle keeps the lengths of all sequences (-uc) and, at the end, the
  // letters of the centroids only (-centroids) - not a copy of the whole read set (1.5 GB on C3)
  C->offs.resize((size_t)nseq + 1);
  for (uint32_t i = 0; i <= nseq; ++i) C->offs[i] = offs[i] - offs[0];
  const char *S = seqs + offs[0]; const uint64_t *O = C->offs.data();
  const bool revcomp = p.strand_both && p.is_nucleo;
  const double t_start = now_s();
  const uint32_t nu = derep_full(S, O, nseq, revcomp, C->seq_unique, C->uniq_seed);
  C->st.s_derep = (float)(now_s() - t_start);
  C->n_unique = nu;
  if (sort_mode != UGS_SORT_NONE && nu) {
    // GetSeqOrder clusterfast.cpp:37-79: QuickSortOrderDesc over the seeds' lengths or DerepResult::GetSumSizeIn (size= with default 1,
    // whether or not -sizein is set: derepresult.cpp:211-225); the uniques are renumbered in that order = the loop order :113-123
    std::vector<uint32_t> v(nu, 0), order(nu), rank(nu), seed2(nu);
    if (sort_mode == UGS_SORT_LENGTH) for (uint32_t u = 0; u < nu; ++u) v[u] = (uint32_t)(O[C->uniq_seed[u] + 1] - O[C->uniq_seed[u]]);
    else for (uint32_t i = 0; i < nseq; ++i) v[C->seq_unique[i]] += (size_in && size_in[i] != 0xffffffffu) ? size_in[i] : 1;
    for (uint32_t u = 0; u < nu; ++u) order[u] = u;
    qs_order_desc_u(v.data(), 0, (int)nu - 1, order.data());
    for (uint32_t k = 0; k < nu; ++k) { rank[order[k]] = k; seed2[k] = C->uniq_seed[order[k]]; }
    for (uint32_t i = 0; i < nseq; ++i) C->seq_unique[i] = rank[C->seq_unique[i]];
    C->uniq_seed.swap(seed2);
  }
  C->uniq_size.assign(nu, 0);
  for (uint32_t i = 0; i < nseq; ++i) C->uniq_size[C->seq_unique[i]] += sizein ? size_in[i] : 1;     // ClusterSink::GetSize clustersink.cpp:119-150
  C->uniq_cluster.assign(nu, 0); C->uniq_nhits.assign(nu, 0); C->uniq_hit_off.assign((size_t)nu + 1, 0);
  uint32_t maxlen = 0;
  for (uint32_t u = 0; u < nu; ++u) maxlen = std::max<uint32_t>(maxlen, (uint32_t)(O[C->uniq_seed[u] + 1] - O[C->uniq_seed[u]]));

  struct Guard { ugs_db *db = nullptr; ugs_batch *b = nullptr; ugs_cluster *c = nullptr; ~Guard() { if (b) ugs_batch_destroy(b); if (db) ugs_db_destroy(db); delete c; } } G;
  G.c = C;
  const uint64_t zero = 0;
  RCCHK(ugs_db_create(&p, "", &zero, 0, device, &G.db));
  ugs_db *db = G.db;
  db->max_tlen = maxlen; db->v.max_tlen = maxlen;           // every centroid is one of the input sequences: plan the kernels for the longest
  if (maxlen >= (uint32_t)p.word_len + 255u) db->gsize_limit = 2048;   // > 255 query words: 16-bit counters on the small path
  const bool profile = getenv("UGS_CLUSTER_PROFILE") != nullptr;       // (debug switches are read once per call, never inside the batch loop)
  uint32_t Bmax = 32768;            // (C3: 16 384 -> 32 768 takes 0.1 s off: fewer, fuller launches; 65 536 loses it again to the in-batch stage)
  if (const char *e = getenv("UGS_CLUSTER_BATCH")) { const int v = atoi(e); if (v >= 1 && v <= (1 << 20)) Bmax = (uint32_t)v; }
  Bmax = std::min<uint32_t>(Bmax, std::max<uint32_t>(nu, 1));
  RCCHK(ugs_batch_create(db, Bmax, (uint64_t)Bmax * maxlen, &G.b));
  G.b->cl_mode = true;
  ugs_batch *b = G.b;
  hipStream_t st = db->stream;
  const uint32_t ns = b->nstrand, K = b->K;
  const uint64_t umax = (uint64_t)Bmax * ns;
  DevBuf d_ucost, d_uorder, d_ohist;
  RCCHK(d_ucost.need(umax * 4)); RCCHK(d_uorder.need(umax * 4)); RCCHK(d_ohist.need(512 * 4));
  DevBuf d_slots, d_elist;
  std::vector<uint2> h_slots, h_over; std::vector<uint32_t> h_elist, h_ooff;
  DevBuf d_ckey, d_clev, d_clinfo, d_walk, d_entn, d_entoff, d_ent, d_pmap, d_pcand, d_pcandn, d_phitn, d_phits, d_pcompact, d_pqn, d_pqoff, d_scan;
  RCCHK(d_ckey.need(umax * K * 8)); RCCHK(d_clev.need(umax * UGS_CL_EV * 8)); RCCHK(d_clinfo.need(umax * 16)); RCCHK(d_walk.need(umax * 4));
  RCCHK(d_entn.need(umax * 4)); RCCHK(d_entoff.need(umax * 4));

  BatchHost H;
  std::vector<char> stage; std::vector<uint64_t> stage_off;
  std::vector<uint8_t> status;
  std::vector<uint32_t> cidx;                         // cluster of an in-batch centroid
  std::vector<uint32_t> pmap, pcand, pcandn, pent0;   // pair units
  uint32_t next = 0, nc = 0;
  uint32_t B_prev = 0; uint64_t pairs_prev = 0;
  const uint32_t Kp = 16;
  std::vector<char> app_seq; std::vector<uint64_t> app_off;

  while (next < nu) {
    // ---- batch size: small while the database is small (the in-batch share of the candidates must stay a minority),
    // never across the small -> Big latch by more than the replay can cut off, halved / doubled on the pair load
    const uint32_t n0 = db->v.nseq;
    uint32_t B = std::max<uint32_t>(64, n0 / 2);
    if (n0 <= p.big) B = std::min<uint64_t>(B, std::max<uint64_t>(64, 4ull * (p.big + 1 - n0)));
    if (B_prev) {
      if (pairs_prev > 6ull * B_prev) B = std::min(B, std::max<uint32_t>(256, B_prev / 2));
      else if (pairs_prev < 2ull * B_prev) B = std::min(B, B_prev * 2); else B = std::min(B, B_prev);
    }
    B = std::min(B, Bmax); B = std::min(B, nu - next);
    // ---- upload the batch (the uniques' seed sequences, as read) and search the frozen index
    double tq = now_s();
    stage.clear(); stage_off.assign(1, 0);
    for (uint32_t k = 0; k < B; ++k) {
      const uint32_t si = C->uniq_seed[next + k];
      stage.insert(stage.end(), S + O[si], S + O[si + 1]);
      stage_off.push_back(stage.size());
    }
    RCCHK(ugs_batch_upload(b, stage.data(), stage_off.data(), B));
    if (profile) fprintf(stderr, "[ugs] batch %u: n0 %u B %u stage+upload %.4f s\n", C->st.batches, n0, B, now_s() - tq);
    const uint32_t units = B * ns;
    b->v.unit_cost = (uint32_t *)d_ucost.p; b->v.unit_order = (uint32_t *)d_uorder.p; b->v.order_hist = (uint32_t *)d_ohist.p;
    b->v.cand_key = (uint64_t *)d_ckey.p; b->v.cl_ev = (uint64_t *)d_clev.p; b->v.cl_info = (uint32_t *)d_clinfo.p; b->v.walk_n = (uint32_t *)d_walk.p;
    b->v.unit_map = nullptr;
    RCCHK(ugs_batch_search(b));
    RCCHK(ugs_batch_sync(b));
    if (profile) fprintf(stderr, "[ugs] batch %u: search done %.4f s\n", C->st.batches, now_s() - tq);
    const bool small_path = !db->v.big;
    C->st.s_search += (float)(now_s() - tq); tq = now_s();
    // ---- the batch's own index and the in-batch word counts (count, scan, write)
    uint64_t *d_brow = nullptr; uint32_t *d_bpost = nullptr; uint64_t n_bpost = 0; uint32_t bmax = 0;
    RCCHK(ugs_build_index(db->d_tab, b->d_qseqs, b->d_qoffs, B, stage.size(), p.word_len, db->v.alpha, db->v.slots, &d_brow, &d_bpost, &n_bpost, &bmax, st));
    struct FreeIdx { uint64_t *a; uint32_t *b; ~FreeIdx() { (void)ugs_free(a); (void)ugs_free(b); } } free_idx{d_brow, d_bpost};
    // (r6) ONE launch settles nearly every unit: its count and its first INBATCH_SLOTS entries; the few units with more entries (reads
    // without a frozen hit among many relatives in their own batch) get a second launch over their list with exact offsets.  Before: a
    // count-only launch, a host scan and a second FULL launch - the kernel ran twice over every unit (0.9 s of C3's GPU time).
    RCCHK(d_slots.need((size_t)units * INBATCH_SLOTS * 8));
    RCCHK(ugs_launch_inbatch(b->v, d_brow, d_bpost, b->rl.ns_max, small_path, (uint32_t)p.max_rejects, db->num_cu, (uint32_t *)d_entn.p, nullptr, (uint2 *)d_slots.p, st,
                             INBATCH_SLOTS, nullptr, 0));
    H.nq = B; H.ns = ns; H.K = K;
    H.ent_n.resize(units); H.ent_off.resize(units);
    h_slots.resize((size_t)units * INBATCH_SLOTS);
    HIPCHK(hipMemcpyAsync(H.ent_n.data(), d_entn.p, (size_t)units * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(h_slots.data(), d_slots.p, (size_t)units * INBATCH_SLOTS * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    uint64_t n_ent = 0, n_over = 0;
    h_elist.clear(); h_ooff.assign(units, 0);
    for (uint32_t u = 0; u < units; ++u) {
      H.ent_off[u] = (uint32_t)n_ent; n_ent += H.ent_n[u];
      if (H.ent_n[u] > INBATCH_SLOTS) { h_elist.push_back(u); h_ooff[u] = (uint32_t)n_over; n_over += H.ent_n[u]; }
    }
    if (n_ent > 0x7fffffffull) { ugs_set_error("in-batch candidate list overflow"); return UGS_E_ENVELOPE; }
    H.ent.resize(n_ent);
    h_over.resize(n_over);
    if (n_over) {
      RCCHK(d_ent.need(n_over * 8));
      RCCHK(d_elist.need(h_elist.size() * 4));
      HIPCHK(hipMemcpyAsync(d_entoff.p, h_ooff.data(), (size_t)units * 4, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(d_elist.p, h_elist.data(), h_elist.size() * 4, hipMemcpyHostToDevice, st));
      RCCHK(ugs_launch_inbatch(b->v, d_brow, d_bpost, b->rl.ns_max, small_path, (uint32_t)p.max_rejects, db->num_cu, (uint32_t *)d_entn.p, (const uint32_t *)d_entoff.p, (uint2 *)d_ent.p, st,
                               0, (const uint32_t *)d_elist.p, (uint32_t)h_elist.size()));
      HIPCHK(hipMemcpyAsync(h_over.data(), d_ent.p, n_over * 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
    }
    for (uint32_t u = 0; u < units; ++u) {
      const uint32_t n = H.ent_n[u];
      if (!n) continue;
      const uint2 *src = n > INBATCH_SLOTS ? h_over.data() + h_ooff[u] : h_slots.data() + (size_t)u * INBATCH_SLOTS;
      memcpy(H.ent.data() + H.ent_off[u], src, (size_t)n * 8);
    }
    HIPCHK(hipStreamSynchronize(st));
    C->st.s_inbatch += (float)(now_s() - tq); tq = now_s();
    // ---- what the frozen search found
    H.cand_key.resize((size_t)units * K); H.cand.resize((size_t)units * K); H.cand_n.resize(units); H.walk_n.resize(units); H.hit_n.resize(units);
    H.cl_info.resize((size_t)units * 4); H.cl_ev.resize((size_t)units * UGS_CL_EV); H.fhits.resize(units);
    HIPCHK(hipMemcpyAsync(H.cand_key.data(), d_ckey.p, (size_t)units * K * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(H.cand.data(), b->d_cand, (size_t)units * K * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(H.cand_n.data(), b->d_cand_n, (size_t)units * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(H.walk_n.data(), d_walk.p, (size_t)units * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(H.hit_n.data(), b->d_hit_n, (size_t)units * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(H.cl_info.data(), d_clinfo.p, (size_t)units * 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(H.cl_ev.data(), d_clev.p, (size_t)units * UGS_CL_EV * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(H.fhits.data(), b->d_hits, (size_t)units * sizeof(ugs_hit), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    H.pair_out.assign(n_ent, 0); H.pair_hit.assign(n_ent, 0);
    C->st.s_d2h += (float)(now_s() - tq); tq = now_s();

    // ---- pass 1 (input order): queries no earlier query of the batch can influence are final at once; for the others
    // every pair that may matter is listed for the device
    status.assign(B, ST_UNKNOWN);
    pmap.clear(); pcand.clear(); pcandn.clear(); pent0.clear();
    uint64_t n_pairs = 0;
    auto frozen_only = [&](uint32_t q) -> int {        // exact result of a query without in-batch influence: ST_MEMBER / ST_CENTROID, or -1 (hard)
      bool hit = false;
      for (uint32_t s = 0; s < ns; ++s) {
        const UnitRes r = replay_unit(H, p, small_path, q * ns + s, status.data(), false);
        if (r.kind == R_HARD || r.kind == R_NEED) return -1;
        hit = hit || r.kind == R_HIT_FROZEN;
      }
      return hit ? ST_MEMBER : ST_CENTROID;
    };
    for (uint32_t q = 0; q < B; ++q) {
      bool any = false;
      for (uint32_t s = 0; s < ns && !any; ++s) any = replay_unit(H, p, small_path, q * ns + s, status.data(), true).kind == R_NEED && H.ent_n[q * ns + s] != 0;
      if (!any) { const int r = frozen_only(q); status[q] = r < 0 ? ST_MAYBE : (uint8_t)r; if (r >= 0) continue; }
      status[q] = ST_MAYBE;
      for (uint32_t s = 0; s < ns; ++s) {
        const uint32_t u = q * ns + s, e0 = H.ent_off[u], en = H.ent_n[u];
        uint32_t cnt = 0;
        for (uint32_t e = e0; e < e0 + en; ++e) {
          if (status[H.ent[e].x] == ST_MEMBER) continue;
          if (cnt % Kp == 0) { pmap.push_back((q << 1) | s); pcandn.push_back(0); pent0.push_back((uint32_t)pcand.size()); pcand.resize(pcand.size() + Kp, 0); }
          pcand[pcand.size() - Kp + (cnt % Kp)] = e;             // entry index for now; the target goes to the device below
          ++pcandn.back(); ++cnt; ++n_pairs;
        }
      }
    }
    C->st.s_replay += (float)(now_s() - tq); tq = now_s();
    // ---- the pair stage: k_align over explicit (query strand, in-batch target) lists; targets are the batch's own letters
    const uint32_t npu = (uint32_t)pmap.size();
    H.phits.clear();
    uint64_t frozen_runs = b->cigar_used_host;
    if (npu) {
      std::vector<uint32_t> pc(pcand.size());
      for (size_t k = 0; k < pcand.size(); ++k) pc[k] = 0;
      for (uint32_t pu = 0; pu < npu; ++pu) for (uint32_t k = 0; k < pcandn[pu]; ++k) pc[(size_t)pu * Kp + k] = H.ent[pcand[(size_t)pu * Kp + k]].x;
      RCCHK(d_pmap.need((size_t)npu * 4)); RCCHK(d_pcand.need((size_t)npu * Kp * 4)); RCCHK(d_pcandn.need((size_t)npu * 4)); RCCHK(d_phitn.need((size_t)npu * 4));
      RCCHK(d_phits.need((size_t)npu * Kp * sizeof(ugs_hit))); RCCHK(d_pcompact.need((size_t)npu * Kp * sizeof(ugs_hit)));
      RCCHK(d_pqn.need((size_t)npu * 4)); RCCHK(d_pqoff.need(((size_t)npu + 1) * 4));
      const size_t scan_bytes = ugs_compact_tmp_bytes(npu);
      RCCHK(d_scan.need(scan_bytes));
      HIPCHK(hipMemcpyAsync(d_pmap.p, pmap.data(), (size_t)npu * 4, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(d_pcand.p, pc.data(), (size_t)npu * Kp * 4, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(d_pcandn.p, pcandn.data(), (size_t)npu * 4, hipMemcpyHostToDevice, st));
      UgsDbView pv = db->v;
      pv.seqs = b->d_qseqs; pv.offs = b->d_qoffs; pv.nseq = B; pv.pk = nullptr;     // (targets = the batch's own letters: bytes only)
      pv.max_accepts = (int32_t)Kp; pv.max_rejects = 0x7fffffff; pv.align_flags |= UGS_A_NOTERM;
      for (int attempt = 0;; ++attempt) {
        // the path pool is shared with the frozen stage's hits: make room for the pairs behind them
        const uint64_t want = frozen_runs + n_pairs * 12 + 4096;
        if (attempt == 0 && want > b->cigar_cap) {
          uint32_t *np_ = nullptr;
          const uint64_t cap = want + want / 4;
          HIPCHK(ugs_malloc(&np_, cap * 4));
          if (frozen_runs) HIPCHK(hipMemcpyAsync(np_, b->d_cigar, frozen_runs * 4, hipMemcpyDeviceToDevice, st));
          HIPCHK(hipStreamSynchronize(st));
          HIPCHK(ugs_free(b->d_cigar)); b->d_cigar = np_; b->cigar_cap = cap; b->v.cigar_pool = np_; b->v.cigar_cap = cap;
        }
        UgsBatchView bv2 = b->v;
        bv2.nq = npu; bv2.nstrand = 1; bv2.K = Kp;
        bv2.cand = (uint32_t *)d_pcand.p; bv2.cand_n = (uint32_t *)d_pcandn.p; bv2.hits = (ugs_hit *)d_phits.p; bv2.hit_n = (uint32_t *)d_phitn.p;
        bv2.unit_map = (const uint32_t *)d_pmap.p; bv2.walk_n = nullptr; bv2.cand_key = nullptr; bv2.cl_ev = nullptr; bv2.cl_info = nullptr;
        bv2.qpk = nullptr;               // (the planes k_rank_setup packed are indexed by the search's units, not by this stage's pairs)
        const unsigned long long fr = frozen_runs;
        HIPCHK(hipMemcpyAsync(b->d_cigar_used, &fr, 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemsetAsync(b->d_ctr + UGS_CTR_NEXT_UNIT, 0, 8, st));
        UgsAlignLaunch al = b->al;
        al.grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((npu + al.wpb - 1) / al.wpb, (uint64_t)b->al.grid));
        RCCHK(ugs_launch_align(pv, bv2, al, st));
        unsigned long long used = 0, err = 0;
        HIPCHK(hipMemcpyAsync(&used, b->d_cigar_used, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&err, b->d_ctr + UGS_CTR_ERR, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (err) { ugs_set_error("device envelope exceeded in the pair stage (flags 0x%llx)", err); return UGS_E_ENVELOPE; }
        if (used <= b->cigar_cap) { b->cigar_used_host = used; break; }
        if (attempt == 2) { ugs_set_error("path pool overflow persisted"); return UGS_E_CAPACITY; }
        uint32_t *np_ = nullptr;
        const uint64_t cap = used + used / 4 + 4096;
        HIPCHK(ugs_malloc(&np_, cap * 4));
        if (frozen_runs) HIPCHK(hipMemcpyAsync(np_, b->d_cigar, frozen_runs * 4, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(ugs_free(b->d_cigar)); b->d_cigar = np_; b->cigar_cap = cap; b->v.cigar_pool = np_; b->v.cigar_cap = cap;
      }
      RCCHK(ugs_compact_hits((const uint32_t *)d_phitn.p, (const ugs_hit *)d_phits.p, npu, 1, Kp, (uint32_t *)d_pqn.p, (uint32_t *)d_pqoff.p,
                             (ugs_hit *)d_pcompact.p, d_scan.p, d_scan.cap, 0, st));
      std::vector<uint32_t> pqn(npu);
      HIPCHK(hipMemcpyAsync(pqn.data(), d_pqn.p, (size_t)npu * 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      uint64_t tot = 0;
      for (uint32_t pu = 0; pu < npu; ++pu) tot += pqn[pu];
      H.phits.resize(tot);
      if (tot) HIPCHK(hipMemcpyAsync(H.phits.data(), d_pcompact.p, tot * sizeof(ugs_hit), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      uint64_t hp = 0;
      for (uint32_t pu = 0; pu < npu; ++pu) {
        uint32_t taken = 0;
        for (uint32_t k = 0; k < pcandn[pu]; ++k) {
          const uint32_t e = pcand[(size_t)pu * Kp + k];
          if (taken < pqn[pu] && H.phits[hp + taken].target == H.ent[e].x) { H.pair_out[e] = 2; H.pair_hit[e] = (uint32_t)(hp + taken); ++taken; }
          else H.pair_out[e] = 1;
        }
        if (taken != pqn[pu]) { ugs_set_error("pair stage bookkeeping mismatch"); return UGS_E_HIP; }
        hp += pqn[pu];
      }
    }
    H.pool.resize(b->cigar_used_host);
    if (b->cigar_used_host) HIPCHK(hipMemcpy(H.pool.data(), b->d_cigar, b->cigar_used_host * 4, hipMemcpyDeviceToHost));

    C->st.s_pairs += (float)(now_s() - tq); tq = now_s();
    // ---- pass 2 (input order): every earlier query of the batch is decided when a query is replayed
    for (uint32_t q = 0; q < B; ++q) if (status[q] == ST_MAYBE) status[q] = ST_UNKNOWN;
    cidx.assign(B, 0);
    app_seq.clear(); app_off.assign(1, 0);
    uint32_t done = 0;
    uint32_t nc_batch = 0;
    for (uint32_t q = 0; q < B; ++q) {
      // the latch: a query that sees more than -big centroids is searched on the Big path (udbusortedsearcher.cpp:43-44)
      if (small_path && (uint64_t)n0 + nc_batch > p.big) break;
      UnitRes r[2]; bool bad = false;
      for (uint32_t s = 0; s < ns; ++s) { r[s] = replay_unit(H, p, small_path, q * ns + s, status.data(), false); if (r[s].kind == R_HARD || r[s].kind == R_NEED) bad = true; }
      if (bad) {
        if (q == 0) { ugs_set_error("cluster_fast: the first query of a batch could not be replayed (prefix maxima %u > %d)", H.cl_info[2], UGS_CL_EV); return UGS_E_ENVELOPE; }
        ++C->st.batches_cut;
        break;
      }
      const uint32_t u = next + q;
      ugs_hit hh[2]; uint32_t nh = 0;
      for (uint32_t s = 0; s < ns; ++s) {
        if (r[s].kind == R_HIT_FROZEN) hh[nh++] = H.fhits[q * ns + s];
        else if (r[s].kind == R_HIT_BATCH) { ugs_hit h = H.phits[H.pair_hit[r[s].idx]]; h.target = cidx[H.ent[r[s].idx].x]; hh[nh++] = h; ++C->st.hits_in_batch; }
      }
      C->uniq_nhits[u] = nh;
      if (nh == 0) {
        status[q] = ST_CENTROID; cidx[q] = nc;
        C->uniq_cluster[u] = nc; C->centroid_uniq.push_back(u); C->cluster_size.push_back(C->uniq_size[u]); ++nc; ++nc_batch;
        app_seq.insert(app_seq.end(), stage.begin() + stage_off[q], stage.begin() + stage_off[q + 1]);
        app_off.push_back(app_seq.size());
      } else {
        status[q] = ST_MEMBER;
        // HitMgr::GetTopHit hitmgr.cpp:400-420: best float(FractId), ties to the smaller target index, else the earlier hit
        uint32_t top = 0; float tops = 0; uint32_t mint = 0;
        for (uint32_t i = 0; i < nh; ++i) {
          const float sc = (float)(hh[i].aln_len == 0 ? 0.0 : (double)hh[i].ids / (double)hh[i].aln_len);
          if (i == 0 || sc > tops || (sc == tops && hh[i].target < mint)) { top = i; tops = sc; mint = hh[i].target; }
        }
        const uint32_t c = hh[top].target;
        C->uniq_cluster[u] = c; C->cluster_size[c] += C->uniq_size[u];
        for (uint32_t i = 0; i < nh; ++i) { hh[i].query = u; hh[i].flags = (hh[i].flags & 0xffu) | (i << UGS_HIT_ORDER_SHIFT); }
        if (nh == 2) {                                                   // HitMgr::Sort hitmgr.cpp:477-483
          float sc[2]; unsigned ord[2] = {0, 1};
          for (uint32_t i = 0; i < 2; ++i) sc[i] = (float)(hh[i].aln_len == 0 ? 0.0 : (double)hh[i].ids / (double)hh[i].aln_len);
          ugs_qs_order_desc(sc, 0, 1, ord);
          if (ord[0] == 1) std::swap(hh[0], hh[1]);
        }
        for (uint32_t i = 0; i < nh; ++i) {
          ugs_hit h = hh[i];
          const uint64_t co = C->pool.size();
          C->pool.insert(C->pool.end(), H.pool.begin() + h.cigar_off, H.pool.begin() + h.cigar_off + h.cigar_len);
          h.cigar_off = co;
          C->hits.push_back(h);
        }
      }
      C->uniq_hit_off[u + 1] = C->hits.size();
      done = q + 1;
    }
    if (done == 0) { ugs_set_error("cluster_fast made no progress (internal error)"); return UGS_E_HIP; }
    C->st.s_replay += (float)(now_s() - tq); tq = now_s();
    if (nc_batch) RCCHK(ugs_db_append(db, app_seq.data(), app_off.data(), nc_batch));
    C->st.s_append += (float)(now_s() - tq);
    ++C->st.batches; C->st.pairs_in_batch += n_pairs; C->st.inbatch_entries += n_ent; C->st.queries_redone += B - done;
    C->st.max_batch = std::max<uint32_t>(C->st.max_batch, B);
    ugs_batch_stats bs;
    if (ugs_batch_get_stats(b, &bs) == UGS_OK) { uint64_t kh[6] = {0, 0, 0, 0, 0, 0}; if (profile) { (void)ugs_debug_kernel_hits(b, kh, 6); fprintf(stderr, "[ugs] batch %u: bitmap kernel %.3f ms, k_rank behind it %.3f ms ; T0..T7 %llu %llu %llu %llu %llu %llu %llu %llu\n", C->st.batches, kh[4] / 1000.0, kh[5] / 1000.0, b->ctr[UGS_CTR_T0], b->ctr[UGS_CTR_T1], b->ctr[UGS_CTR_T2], b->ctr[UGS_CTR_T3], b->ctr[UGS_CTR_T4], b->ctr[UGS_CTR_T5], b->ctr[UGS_CTR_T6], b->ctr[UGS_CTR_T7]); } if (profile) fprintf(stderr, "[ugs] batch %u: n0 %u B %u done %u path %s ms_setup %.3f ms_rank %.3f ms_align %.3f pairs %llu | bitmap kernel: %s units %llu deferred %llu\n", C->st.batches, n0, B, done, small_path ? "small" : "big", bs.ms_rank_setup, bs.ms_rank, bs.ms_align, (unsigned long long)n_pairs, b->r2_ran ? "ran" : "-", b->ctr[UGS_CTR_R2_DONE], b->ctr[UGS_CTR_DEFER]); C->st.units_heavy += (uint32_t)b->ctr[UGS_CTR_HV_DONE]; if (profile) fprintf(stderr, "[ugs] batch %u: heavy-unit kernel ranked %llu of the deferred units, %llu went on to k_rank\n", C->st.batches, b->ctr[UGS_CTR_HV_DONE], b->ctr[UGS_CTR_DEFER2]);
      C->st.ms_rank += bs.ms_rank + bs.ms_rank_setup; C->st.ms_align += bs.ms_align; C->st.postings += bs.postings; C->st.pairs_frozen += bs.pairs_aligned; }
    next += done;
    B_prev = B; pairs_prev = n_pairs;
  }
  C->n_clusters = nc;
  C->cent_off.assign((size_t)nc + 1, 0);
  for (uint32_t c = 0; c < nc; ++c) { const uint32_t si = C->uniq_seed[C->centroid_uniq[c]]; C->cent_off[c + 1] = C->cent_off[c] + (O[si + 1] - O[si]); }
  C->seqs.resize(C->cent_off[nc]);
  for (uint32_t c = 0; c < nc; ++c) { const uint32_t si = C->uniq_seed[C->centroid_uniq[c]]; memcpy(C->seqs.data() + C->cent_off[c], S + O[si], O[si + 1] - O[si]); }
  C->st.s_total = (float)(now_s() - t_start);
  G.c = nullptr;
  *out = C;
  return UGS_OK;
}

extern "C" int ugs_cluster_counts(const ugs_cluster *c, uint32_t *n_unique, uint32_t *n_clusters, uint64_t *n_hits, uint64_t *cigar_runs)
{
  if (!c) return UGS_E_ARG;
  if (n_unique) *n_unique = c->n_unique;
  if (n_clusters) *n_clusters = c->n_clusters;
  if (n_hits) *n_hits = c->hits.size();
  if (cigar_runs) *cigar_runs = c->pool.size();
  return UGS_OK;
}

extern "C" int ugs_cluster_get(const ugs_cluster *c, uint32_t *seq_unique, uint32_t *uniq_seed, uint32_t *uniq_cluster, uint32_t *uniq_nhits,
                               uint32_t *centroid_uniq, uint32_t *cluster_size, ugs_hit *hits, uint32_t *cigar_pool)
{
  if (!c) return UGS_E_ARG;
  if (seq_unique) memcpy(seq_unique, c->seq_unique.data(), (size_t)c->nseq * 4);
  if (uniq_seed) memcpy(uniq_seed, c->uniq_seed.data(), (size_t)c->n_unique * 4);
  if (uniq_cluster) memcpy(uniq_cluster, c->uniq_cluster.data(), (size_t)c->n_unique * 4);
  if (uniq_nhits) memcpy(uniq_nhits, c->uniq_nhits.data(), (size_t)c->n_unique * 4);
  if (centroid_uniq) memcpy(centroid_uniq, c->centroid_uniq.data(), (size_t)c->n_clusters * 4);
  if (cluster_size) memcpy(cluster_size, c->cluster_size.data(), (size_t)c->n_clusters * 4);
  if (hits && !c->hits.empty()) memcpy(hits, c->hits.data(), c->hits.size() * sizeof(ugs_hit));
  if (cigar_pool && !c->pool.empty()) memcpy(cigar_pool, c->pool.data(), c->pool.size() * 4);
  return UGS_OK;
}

extern "C" int ugs_cluster_get_stats(const ugs_cluster *c, ugs_cluster_stats *st)
{
  if (!c || !st) return UGS_E_ARG;
  *st = c->st;
  return UGS_OK;
}

// labels: nseq NUL-terminated strings, concatenated in input order
static bool split_labels(const char *labels, uint32_t n, std::vector<const char *> &out)
{
  out.resize(n);
  const char *p = labels;
  for (uint32_t i = 0; i < n; ++i) { out[i] = p; p += strlen(p) + 1; }
  return true;
}

// -uc of cluster_fast: per unique in input order its S record (OutputUCNoHits outputuc.cpp:10-43) or H records
// (OutputUC :45-93), each followed by the same record for the unique's duplicates; then the C records
// (ClusterSink::WriteUC_CRecs clustersink.cpp:477-493)
extern "C" int ugs_cluster_write_uc(const ugs_cluster *c, const char *labels, const char *path)
{
  if (!c || !labels || !path) { ugs_set_error("null argument"); return UGS_E_ARG; }
  std::vector<const char *> lab;
  split_labels(labels, c->nseq, lab);
  FILE *f = fopen(path, "wb");
  if (!f) { ugs_set_error("cannot create %s", path); return UGS_E_ARG; }
  std::vector<std::vector<uint32_t>> dup(c->n_unique);            // members beyond the seed, in input order
  for (uint32_t i = 0; i < c->nseq; ++i) if (c->uniq_seed[c->seq_unique[i]] != i) dup[c->seq_unique[i]].push_back(i);
  std::vector<char> buf(1 << 16);
  bool ok = true;
  for (uint32_t u = 0; u < c->n_unique && ok; ++u) {
    const uint32_t seed = c->uniq_seed[u];
    const uint32_t L = (uint32_t)(c->offs[seed + 1] - c->offs[seed]);
    if (c->uniq_nhits[u] == 0) {
      const uint32_t ci = c->uniq_cluster[u];
      ok = ok && fprintf(f, "S\t%u\t%u\t*\t.\t*\t*\t*\t%s\t*\n", ci, L, lab[seed]) >= 0;
      for (uint32_t m : dup[u]) ok = ok && fprintf(f, "H\t%u\t%u\t100.0\t.\t0\t%u\t=\t%s\t%s\n", ci, L, L, lab[m], lab[seed]) >= 0;
      continue;
    }
    for (uint64_t k = c->uniq_hit_off[u]; k < c->uniq_hit_off[u + 1]; ++k) {
      const ugs_hit &h = c->hits[k];
      const char *tl = lab[c->uniq_seed[c->centroid_uniq[h.target]]];
      for (int pass = 0; pass < 1 + (int)dup[u].size(); ++pass) {
        const char *ql = pass == 0 ? lab[seed] : lab[dup[u][pass - 1]];
        int n = ugs_format_uc_hit(&h, c->pool.data(), c->p.is_nucleo, ql, tl, buf.data(), (int)buf.size());
        if (n >= (int)buf.size()) { buf.resize((size_t)n + 16); n = ugs_format_uc_hit(&h, c->pool.data(), c->p.is_nucleo, ql, tl, buf.data(), (int)buf.size()); }
        if (n < 0) { ok = false; break; }                          // formatter error: nothing sensible to write
        ok = ok && fwrite(buf.data(), 1, (size_t)n, f) == (size_t)n;
      }
    }
  }
  for (uint32_t ci = 0; ci < c->n_clusters && ok; ++ci)
    ok = ok && fprintf(f, "C\t%u\t%u\t*\t*\t*\t*\t*\t%s\t*\n", ci, c->cluster_size[ci], lab[c->uniq_seed[c->centroid_uniq[ci]]]) >= 0;
  if (fclose(f) != 0) ok = false;
  if (!ok) { ugs_set_error("write error on %s", path); return UGS_E_ARG; }
  return UGS_OK;
}

// -centroids (ClusterSink::CentroidsToFASTA clustersink.cpp:262-289): centroids by decreasing cluster size in the order of the
// reference's QuickSortOrderDesc, 80 columns (SeqToFasta seqdb.cpp:62-90)
extern "C" int ugs_cluster_write_centroids(const ugs_cluster *c, const char *labels, const char *path)
{
  return ugs_cluster_write_centroids_sized(c, labels, path, 0, 0);
}

// StripSize = StripAnnot(Label, "size=") label.cpp:46-71 over Split(';') myutils.cpp:1588-1607
static void strip_size(std::string &label)
{
  if (label.find("size=") == std::string::npos) return;
  std::string out, field;
  auto put = [&](const std::string &f) { if (f.compare(0, 5, "size=") != 0) { out += f; out += ';'; } };
  for (char ch : label) { if (ch == ';') { put(field); field.clear(); } else field.push_back(ch); }
  if (!field.empty()) put(field);
  if (out.find('=') == std::string::npos && !out.empty()) out.pop_back();
  label.swap(out);
}

// AppendSize -> Psasc myutils.cpp:824-839
static void append_size(std::string &label, uint32_t size)
{
  if (!label.empty() && label.back() != ';') label += ';';
  label += "size=" + std::to_string(size) + ";";
}

// + MakeCentroidLabel clustersink.cpp:219-243 (-sizein | -sizeout strip size=, -sizeout appends the cluster size) and -minsize (:275-277)
extern "C" int ugs_cluster_write_centroids_sized(const ugs_cluster *c, const char *labels, const char *path, int size_flags, uint32_t minsize)
{
  if (!c || !labels || !path) { ugs_set_error("null argument"); return UGS_E_ARG; }
  std::vector<const char *> lab;
  split_labels(labels, c->nseq, lab);
  FILE *f = fopen(path, "wb");
  if (!f) { ugs_set_error("cannot create %s", path); return UGS_E_ARG; }
  std::vector<uint32_t> order(c->n_clusters);
  for (uint32_t i = 0; i < c->n_clusters; ++i) order[i] = i;
  if (c->n_clusters) qs_order_desc_u(c->cluster_size.data(), 0, (int)c->n_clusters - 1, order.data());
  bool ok = true;
  for (uint32_t k = 0; k < c->n_clusters && ok; ++k) {
    const uint32_t si = c->uniq_seed[c->centroid_uniq[order[k]]];
    const char *s = c->seqs.data() + c->cent_off[order[k]];
    const uint32_t L = (uint32_t)(c->cent_off[order[k] + 1] - c->cent_off[order[k]]);
    if (c->cluster_size[order[k]] < minsize) break;
    if (L == 0) continue;
    std::string label = lab[si];
    if (size_flags & (UGS_SIZEIN | UGS_SIZEOUT)) strip_size(label);
    if (size_flags & UGS_SIZEOUT) append_size(label, c->cluster_size[order[k]]);
    ok = ok && fprintf(f, ">%s\n", label.c_str()) >= 0;
    for (uint32_t from = 0; from < L && ok; from += 80) {
      const uint32_t n = std::min<uint32_t>(80, L - from);
      ok = ok && fwrite(s + from, 1, n, f) == n && fputc('\n', f) != EOF;
    }
  }
  if (fclose(f) != 0) ok = false;
  if (!ok) { ugs_set_error("write error on %s", path); return UGS_E_ARG; }
  return UGS_OK;
}
