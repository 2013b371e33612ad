// ugs_rank2.h - launch interface of the dense-index Big-path ranking kernel (ugs_rank2.hip).  Internal.
#pragma once
#include "ugs_dev.h"

struct UgsRank2Params {
  uint32_t ns_max;     // stride of UgsBatchView::unit_slots
  uint32_t G;          // targets per partition: a multiple of 8192, <= 65536 (one bit per target in LDS)
  uint32_t np;         // partitions = ceil(nseq / G) = UgsDbView::np2
  uint32_t kcap;       // kept keys per unit the LDS list holds (more defers the unit to k_rank)
  uint32_t W;          // partitions per window of the scan (a multiple of 4); k_rank3g: partitions per super-partition (0: per unit)
  uint32_t clcap;      // chunk descriptors the LDS list of a window holds; k_rank3g: postings per super-partition aimed at (W == 0)
  uint32_t lds;        // dynamic LDS bytes per wave
  const uint16_t *post16; // k_rank2 only, null otherwise: the postings as 16-bit offsets inside their partition (target mod G) at the SAME element
                       //    positions as UgsDbView::postings - the kernel then streams these (np <= 512)
  uint32_t *defer2;    // cluster_fast only, null otherwise: behind the CL instantiation the heavy-unit instantiation (HV) takes the deferred units
                       //    and lists the ones IT cannot take here (counters[UGS_CTR_DEFER2] of them) for k_rank
  uint32_t hv_lds;     // its dynamic LDS bytes per wave, hv_grid its grid (0: no such stage)
  int hv_grid;
  uint32_t force_defer; // tests (UGS_R2_HV=2): the CL instantiation defers EVERY unit, so that the heavy-unit kernel ranks a whole batch
  uint32_t gather;     // 1: k_rank2g (sparse index: one chunk = the sub-rows of all sampled rows of a partition); 2: k_rank3g (sparse index,
                       //    two filter passes per super-partition, ugs_rank3.hip)
};

size_t ugs_rank2_lds(uint32_t G, uint32_t kcap, uint32_t clcap, int cl = 0);
size_t ugs_rank2_hv_lds(uint32_t gsize, uint32_t kcap, uint32_t clcap);      // the heavy-unit instantiation: 4-bit counters for gsize targets
size_t ugs_rank2g_lds(uint32_t G, uint32_t kcap, uint32_t np);
int ugs_rank2_blocks_per_cu(size_t lds, int gather, int cl = 0, int p16 = 0, int hv = 0);      // cl: the cluster_fast instantiation (walk records); p16: 16-bit postings
int ugs_build_post16(const uint32_t *d_postings, uint64_t n, uint32_t G, uint16_t *d_out, hipStream_t st);   // out[i] = postings[i] mod G
// k_rank3g (ugs_rank3.hip)
size_t ugs_rank3g_lds(uint32_t kcap);
int ugs_rank3g_blocks_per_cu(size_t lds);
int ugs_launch_rank3g(const UgsDbView &db, const UgsBatchView &b, const UgsRank2Params &prm, int grid, hipStream_t st);
int ugs_launch_rank2(const UgsDbView &db, const UgsBatchView &b, const UgsRank2Params &prm, int grid, hipStream_t st);
