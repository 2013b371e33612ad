// ugs_gather.cpp - include/ugs_comm.h: the gather of the ranks' device-resident hit tables to one rank over RCCL
// (BASELINE config C4).  Built into libugs_rccl.so (links librccl + libugs); libugs.so itself has no RCCL
// dependency.  The tables are moved as bytes; every rank's path offsets are rebased to the concatenated pool by a kernel over the
// staged table on the destination (k_rebase_paths), so that rank 0's host does no per-hit work before HitMgr::Sort.
#include "ugs_host.h"
#include "../../include/ugs_comm.h"
#include <rccl/rccl.h>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <vector>

#define NCCLCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { ugs_set_error("%s:%d %s: %s", __FILE__, __LINE__, #x, ncclGetErrorString(r_)); return UGS_E_HIP; } } while (0)

namespace {
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// the loopback transport's shared state: a generation barrier + a board where the ranks post sizes and device pointers
struct Loop {
  int world;
  std::mutex m; std::condition_variable cv; int arrived = 0; uint64_t gen = 0;
  std::vector<uint64_t> sizes; std::vector<const void *> ptrs;
  explicit Loop(int w) : world(w), sizes((size_t)w * 4, 0), ptrs((size_t)w * 3, nullptr) {}
  void barrier() {
    std::unique_lock<std::mutex> l(m);
    const uint64_t g = gen;
    if (++arrived == world) { arrived = 0; ++gen; cv.notify_all(); }
    else cv.wait(l, [&] { return gen != g; });
  }
};
}  // namespace

struct ugs_comm {
  int rank = 0, world = 1, device = 0;
  ncclComm_t nccl = nullptr;
  std::shared_ptr<Loop> loop;
  hipStream_t st = nullptr;
  uint64_t *d_my = nullptr, *d_all = nullptr;            // 3 sizes + a status word of this rank / of every rank (device, for ncclAllGather)
  std::vector<uint64_t> all4;                             // host copy of the last exchange: [world][4]
  void *d_stage[3] = {nullptr, nullptr, nullptr}; uint64_t stage_cap[3] = {0, 0, 0};     // dst: the gathered tables
  // the last gather (dst)
  std::vector<uint64_t> all;                              // [world][3] bytes of hits / counts / pool
  bool have_last = false; int last_local = 0; bool last_sort = false;
  double s_exchange = 0, s_fetch = 0;
};

// hits [0, n) of one source rank in the staged table: their paths start `base` runs further into the concatenated pool
__global__ void k_rebase_paths(ugs_hit *hits, uint64_t n, uint64_t base)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) hits[i].cigar_off += base;
}

static int comm_common_init(ugs_comm *c)
{
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking));
  HIPCHK(ugs_malloc(&c->d_my, 4 * 8));
  HIPCHK(ugs_malloc(&c->d_all, (size_t)c->world * 4 * 8));
  c->all.assign((size_t)c->world * 3, 0);
  c->all4.assign((size_t)c->world * 4, 0);
  return UGS_OK;
}

extern "C" int ugs_comm_unique_id(char id[UGS_COMM_ID_BYTES])
{
  static_assert(sizeof(ncclUniqueId) <= UGS_COMM_ID_BYTES, "id size");
  if (!id) return UGS_E_ARG;
  ncclUniqueId u;
  NCCLCHK(ncclGetUniqueId(&u));
  memset(id, 0, UGS_COMM_ID_BYTES);
  memcpy(id, &u, sizeof u);
  return UGS_OK;
}

extern "C" int ugs_comm_init_rank(const char id[UGS_COMM_ID_BYTES], int rank, int world, int device, ugs_comm **out)
{
  if (!id || !out || world < 1 || rank < 0 || rank >= world) { ugs_set_error("ugs_comm_init_rank: bad argument"); return UGS_E_ARG; }
  std::unique_ptr<ugs_comm> c(new ugs_comm());
  c->rank = rank; c->world = world; c->device = device;
  RCCHK(comm_common_init(c.get()));
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  NCCLCHK(ncclCommInitRank(&c->nccl, world, u, rank));
  *out = c.release();
  return UGS_OK;
}

extern "C" int ugs_comm_init_all(int ndev, const int *devices, ugs_comm **out)
{
  if (ndev < 1 || !devices || !out) { ugs_set_error("ugs_comm_init_all: bad argument"); return UGS_E_ARG; }
  std::vector<ncclComm_t> comms((size_t)ndev);
  NCCLCHK(ncclCommInitAll(comms.data(), ndev, devices));
  for (int i = 0; i < ndev; ++i) {
    ugs_comm *c = new ugs_comm();
    c->rank = i; c->world = ndev; c->device = devices[i]; c->nccl = comms[(size_t)i];
    const int rc = comm_common_init(c);
    if (rc != UGS_OK) { delete c; return rc; }
    out[i] = c;
  }
  return UGS_OK;
}

extern "C" int ugs_comm_init_loopback(int world, int device, ugs_comm **out)
{
  if (world < 1 || !out) { ugs_set_error("ugs_comm_init_loopback: bad argument"); return UGS_E_ARG; }
  auto loop = std::make_shared<Loop>(world);
  for (int i = 0; i < world; ++i) {
    ugs_comm *c = new ugs_comm();
    c->rank = i; c->world = world; c->device = device; c->loop = loop;
    const int rc = comm_common_init(c);
    if (rc != UGS_OK) { delete c; return rc; }
    out[i] = c;
  }
  return UGS_OK;
}

extern "C" void ugs_comm_destroy(ugs_comm *c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->st) (void)hipStreamSynchronize(c->st);
  (void)hipDeviceSynchronize();        // (as ugs_db_destroy: nothing in flight on the device when a stream goes)
  if (c->nccl) ncclCommDestroy(c->nccl);
  for (int k = 0; k < 3; ++k) if (c->d_stage[k]) (void)ugs_free(c->d_stage[k]);
  if (c->d_my) (void)ugs_free(c->d_my);
  if (c->d_all) (void)ugs_free(c->d_all);
  if (c->st) (void)hipStreamDestroy(c->st);
  delete c;
}

extern "C" int ugs_comm_rank(const ugs_comm *c) { return c ? c->rank : -1; }
extern "C" int ugs_comm_world(const ugs_comm *c) { return c ? c->world : 0; }

extern "C" int ugs_gather_refetch(ugs_comm *c, ugs_hit *hits, uint64_t hits_cap, uint32_t *nhits_per_query, uint64_t nq_cap,
                                  uint32_t *cigar_pool, uint64_t cigar_cap, uint64_t *n_hits, uint64_t *nq_total, uint64_t *cigar_used)
{
  if (!c || !c->have_last) { ugs_set_error("ugs_gather_refetch: no gathered tables on this rank"); return UGS_E_ARG; }
  HIPCHK(hipSetDevice(c->device));
  uint64_t tot[3] = {0, 0, 0};
  for (int r = 0; r < c->world; ++r) for (int k = 0; k < 3; ++k) tot[k] += c->all[(size_t)r * 3 + k];
  const uint64_t nh = tot[0] / sizeof(ugs_hit), nq = tot[1] / 4, nr = tot[2] / 4;
  if (n_hits) *n_hits = nh;
  if (nq_total) *nq_total = nq;
  if (cigar_used) *cigar_used = nr;
  if (nh > hits_cap || nq > nq_cap || nr > cigar_cap || (nh && !hits) || (nq && !nhits_per_query) || (nr && !cigar_pool)) {
    ugs_set_error("gather: output buffers too small (%llu hits, %llu queries, %llu runs)", (unsigned long long)nh, (unsigned long long)nq, (unsigned long long)nr);
    return UGS_E_CAPACITY;
  }
  const double t0 = now_s();
  if (tot[0]) HIPCHK(hipMemcpyAsync(hits, c->d_stage[0], tot[0], hipMemcpyDeviceToHost, c->st));
  if (tot[1]) HIPCHK(hipMemcpyAsync(nhits_per_query, c->d_stage[1], tot[1], hipMemcpyDeviceToHost, c->st));
  if (tot[2]) HIPCHK(hipMemcpyAsync(cigar_pool, c->d_stage[2], tot[2], hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  c->s_fetch = now_s() - t0;
  if (c->last_sort && nq) return ugs_hits_sort(hits, nhits_per_query, (uint32_t)nq, c->last_local);     // HitMgr::Sort, as ugs_batch_fetch does
  return UGS_OK;
}

extern "C" int ugs_gather_results(ugs_comm *c, ugs_batch *b, uint32_t query_base, int dst,
                                  ugs_hit *hits, uint64_t hits_cap, uint32_t *nhits_per_query, uint64_t nq_cap,
                                  uint32_t *cigar_pool, uint64_t cigar_cap, uint64_t *n_hits, uint64_t *nq_total, uint64_t *cigar_used)
{
  if (!c || !b || dst < 0 || dst >= c->world) { ugs_set_error("ugs_gather_results: bad argument"); return UGS_E_ARG; }
  if (b->db->device != c->device) { ugs_set_error("ugs_gather_results: the batch lives on device %d, the communicator on %d", b->db->device, c->device); return UGS_E_ARG; }
  HIPCHK(hipSetDevice(c->device));
  if (n_hits) *n_hits = 0;
  if (nq_total) *nq_total = 0;
  if (cigar_used) *cigar_used = 0;
  void *src[3] = {nullptr, nullptr, nullptr};
  uint64_t my[4] = {0, 0, 0, 0};                            // three table sizes + this rank's status
  // A rank that fails before the transfer must not leave its peers waiting in it: every rank always takes part in the two
  // small exchanges below (sizes + status, then "dst has room"), and all ranks skip the transfer when any rank reported a failure.
  const int rc_local = ugs_batch_device_results(b, query_base, &src[0], &my[0], &src[1], &my[1], &src[2], &my[2]);
  if (rc_local != UGS_OK) { my[0] = my[1] = my[2] = 0; my[3] = (uint64_t)(uint32_t)(-rc_local); }
  const double t0 = now_s();
  const int W = c->world, R = c->rank;
  auto exchange = [&]() -> int {                             // my[0..3] of every rank -> c->all4 on every rank
    if (c->loop) {
      for (int k = 0; k < 4; ++k) c->loop->sizes[(size_t)R * 4 + k] = my[k];
      for (int k = 0; k < 3; ++k) c->loop->ptrs[(size_t)R * 3 + k] = src[k];
      c->loop->barrier();
      c->all4 = c->loop->sizes;
      c->loop->barrier();                                    // everybody has read the board before anybody posts again
    } else {
      HIPCHK(hipMemcpyAsync(c->d_my, my, 4 * 8, hipMemcpyHostToDevice, c->st));
      NCCLCHK(ncclAllGather(c->d_my, c->d_all, 4, ncclUint64, c->nccl, c->st));
      HIPCHK(hipMemcpyAsync(c->all4.data(), c->d_all, (size_t)W * 4 * 8, hipMemcpyDeviceToHost, c->st));
      HIPCHK(hipStreamSynchronize(c->st));
    }
    return UGS_OK;
  };
  auto failed_rank = [&]() -> int { for (int r = 0; r < W; ++r) if (c->all4[(size_t)r * 4 + 3]) return r; return -1; };
  auto report = [&](int r, const char *where) -> int {
    const int code = -(int)(uint32_t)c->all4[(size_t)r * 4 + 3];
    if (r != R) ugs_set_error("gather: rank %d failed %s (code %d); no table was exchanged", r, where, code);
    return r == R ? code : UGS_E_HIP;
  };
  // ---- everybody learns everybody's table sizes (and whether everybody has tables at all)
  RCCHK(exchange());
  c->have_last = false;
  int bad = failed_rank();
  if (bad >= 0) return report(bad, "before the exchange");
  for (int r = 0; r < W; ++r) for (int k = 0; k < 3; ++k) c->all[(size_t)r * 3 + k] = c->all4[(size_t)r * 4 + k];
  // ---- dst: room for the concatenated tables; its verdict travels in the second exchange
  uint64_t tot[3] = {0, 0, 0};
  for (int r = 0; r < W; ++r) for (int k = 0; k < 3; ++k) tot[k] += c->all[(size_t)r * 3 + k];
  if (R == dst)
    for (int k = 0; k < 3 && !my[3]; ++k)
      if (tot[k] > c->stage_cap[k]) {
        if (c->d_stage[k]) (void)ugs_free(c->d_stage[k]);
        c->d_stage[k] = nullptr; c->stage_cap[k] = 0;
        const uint64_t want = tot[k] + tot[k] / 4 + 4096;
        const hipError_t e = ugs_malloc(&c->d_stage[k], want);
        if (e != hipSuccess) { (void)hipGetLastError(); ugs_set_error("gather: %llu bytes of staging on rank %d: %s", (unsigned long long)want, R, hipGetErrorString(e)); my[3] = (uint64_t)(uint32_t)(-UGS_E_NOMEM); }
        else c->stage_cap[k] = want;
      }
  RCCHK(exchange());
  bad = failed_rank();
  if (bad >= 0) return report(bad, "allocating the staging buffers");
  // ---- the exchange: one grouped set of point-to-point transfers (rank r's table k lands behind the tables of ranks < r)
  if (c->loop) {
    if (R == dst)
      for (int k = 0; k < 3; ++k) {
        uint64_t off = 0;
        for (int r = 0; r < W; ++r) {
          const uint64_t n = c->all[(size_t)r * 3 + k];
          if (n) HIPCHK(hipMemcpyAsync((char *)c->d_stage[k] + off, c->loop->ptrs[(size_t)r * 3 + k], n, hipMemcpyDeviceToDevice, c->st));
          off += n;
        }
      }
    HIPCHK(hipStreamSynchronize(c->st));
    c->loop->barrier();                         // the sources stay untouched until dst has copied them
  } else {
    // (an error inside the group is remembered and reported after ncclGroupEnd: the group is never left open)
    NCCLCHK(ncclGroupStart());
    ncclResult_t gerr = ncclSuccess; hipError_t herr = hipSuccess;
    for (int k = 0; k < 3 && gerr == ncclSuccess && herr == hipSuccess; ++k) {
      if (R == dst) {
        uint64_t off = 0;
        for (int r = 0; r < W && gerr == ncclSuccess && herr == hipSuccess; ++r) {
          const uint64_t n = c->all[(size_t)r * 3 + k];
          if (n) {
            if (r == R) herr = hipMemcpyAsync((char *)c->d_stage[k] + off, src[k], n, hipMemcpyDeviceToDevice, c->st);
            else gerr = ncclRecv((char *)c->d_stage[k] + off, n, ncclUint8, r, c->nccl, c->st);
          }
          off += n;
        }
      } else if (my[k]) gerr = ncclSend(src[k], my[k], ncclUint8, dst, c->nccl, c->st);
    }
    const ncclResult_t eerr = ncclGroupEnd();
    if (herr != hipSuccess) { ugs_set_error("gather: device copy failed: %s", hipGetErrorString(herr)); return UGS_E_HIP; }
    if (gerr != ncclSuccess || eerr != ncclSuccess) { ugs_set_error("gather: RCCL exchange failed: %s", ncclGetErrorString(gerr != ncclSuccess ? gerr : eerr)); return UGS_E_HIP; }
    HIPCHK(hipStreamSynchronize(c->st));
  }
  if (R == dst) {
    // every rank's path offsets point into its own pool: rebase them to the concatenated one - ONCE, on the staged table (a later
    // ugs_gather_refetch copies the same staged bytes again)
    uint64_t h0 = 0, pool_base = 0;
    for (int r = 0; r < W; ++r) {
      const uint64_t n = c->all[(size_t)r * 3] / sizeof(ugs_hit);
      if (pool_base && n) {
        k_rebase_paths<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->st>>>((ugs_hit *)c->d_stage[0] + h0, n, pool_base);
        HIPCHK(hipGetLastError());
      }
      h0 += n; pool_base += c->all[(size_t)r * 3 + 2] / 4;
    }
  }
  c->s_exchange = now_s() - t0;
  c->s_fetch = 0;
  c->have_last = R == dst;
  if (R != dst) return UGS_OK;
  c->last_local = b->db->p.local;
  c->last_sort = b->hit_slots > 1 || b->nstrand > 1;
  return ugs_gather_refetch(c, hits, hits_cap, nhits_per_query, nq_cap, cigar_pool, cigar_cap, n_hits, nq_total, cigar_used);
}

extern "C" int ugs_gather_last_times(const ugs_comm *c, double *s_exchange, double *s_fetch)
{
  if (!c) return UGS_E_ARG;
  if (s_exchange) *s_exchange = c->s_exchange;
  if (s_fetch) *s_fetch = c->s_fetch;
  return UGS_OK;
}
