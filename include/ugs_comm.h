/*
 * ugs_comm.h - the multi-GPU exchange of the usearch_global path (BASELINE config C4): queries are
 * independent and the database is read-only, so every GPU searches its own query shard against its own replica of the
 * index and the ONLY exchange is a gather of the device-resident hit tables (ugs_batch_device_results) to one rank,
 * over RCCL (xGMI between the GPUs of a node).  C-ABI, implemented in libugs_rccl.so (links librccl and libugs).
 *
 * The reference has no counterpart: its parallelism is one searcher object per CPU thread over one shared index
 * (search.cpp:89-141), the outputs are serialised by a lock around each query's sinks (hitmgr.cpp:173-210).  Here the
 * rank that receives the tables plays that role: it gets every rank's hits in rank order = query order.
 */
#ifndef UGS_COMM_H
#define UGS_COMM_H
#include "ugs.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ugs_comm ugs_comm;
#define UGS_COMM_ID_BYTES 128

/* One process per GPU (the launch bench.py uses): rank 0 makes the id (ncclGetUniqueId), passes it to the other ranks by
 * any out-of-band channel (a file, a pipe, MPI ...), and every rank creates its communicator on its device. */
int ugs_comm_unique_id(char id[UGS_COMM_ID_BYTES]);
int ugs_comm_init_rank(const char id[UGS_COMM_ID_BYTES], int rank, int world, int device, ugs_comm **out);
/* One process driving several GPUs with one host thread per device (what ugs_cli -gpus N does): ncclCommInitAll.
 * out[i] is rank i on devices[i]; each communicator is then used by its own thread. */
int ugs_comm_init_all(int ndev, const int *devices, ugs_comm **out);
/* Test transport: `world` ranks inside one process that all sit on ONE device and exchange by device-to-device copies instead
 * of RCCL (RCCL refuses two ranks on one GPU).  Exercises everything of ugs_gather_results except the RCCL calls themselves
 * on a single-GPU box; each rank must be driven by its own host thread. */
int ugs_comm_init_loopback(int world, int device, ugs_comm **out);
void ugs_comm_destroy(ugs_comm *c);
int ugs_comm_rank(const ugs_comm *c);
int ugs_comm_world(const ugs_comm *c);

/*
 * Collective over all ranks of `c`, after ugs_batch_sync(b) on every rank (a rank without queries passes a batch uploaded with
 * nq = 0): the ranks' hit tables travel GPU to GPU to rank `dst`, which receives
 *   hits[n_hits]              all ranks' records in rank order, .query = query_base of the owning rank + index in its batch,
 *                             cigar_off rebased to the concatenated pool; each query's hits in HitMgr::Sort order
 *                             (hitmgr.cpp:477-483) exactly as ugs_batch_fetch returns them
 *   nhits_per_query[nq_total] in rank order
 *   cigar_pool[cigar_used]
 * On the other ranks the output pointers are not touched (may be NULL) and the counts come back 0.
 * UGS_E_CAPACITY on dst when a buffer is too small (the three demands are then in n_hits / nq_total / cigar_used); the
 * collective itself has completed on every rank in that case, so nobody hangs: call ugs_gather_refetch with larger buffers.
 * A rank that fails before the transfer (its batch has no synced search, dst cannot allocate its staging buffers) does not
 * strand its peers: every rank always takes part in two small status exchanges, and when any rank reported a failure ALL
 * ranks return an error (the failing rank its own code, the others UGS_E_HIP naming that rank) without moving a table.
 * Only a failure of the RCCL / HIP calls of the exchange itself leaves the communicator unusable.
 */
int ugs_gather_results(ugs_comm *c, ugs_batch *b, uint32_t query_base, int dst,
                       ugs_hit *hits, uint64_t hits_cap, uint32_t *nhits_per_query, uint64_t nq_cap,
                       uint32_t *cigar_pool, uint64_t cigar_cap, uint64_t *n_hits, uint64_t *nq_total, uint64_t *cigar_used);
/* dst only: copy the tables of the last gather (still in dst's device staging buffers) to the host again */
int ugs_gather_refetch(ugs_comm *c, ugs_hit *hits, uint64_t hits_cap, uint32_t *nhits_per_query, uint64_t nq_cap,
                       uint32_t *cigar_pool, uint64_t cigar_cap, uint64_t *n_hits, uint64_t *nq_total, uint64_t *cigar_used);
/* seconds the last gather spent in the GPU-to-GPU exchange / in the device-to-host copies (dst) */
int ugs_gather_last_times(const ugs_comm *c, double *s_exchange, double *s_fetch);

#ifdef __cplusplus
}
#endif
#endif /* UGS_COMM_H */
