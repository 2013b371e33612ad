/*
 * ugs_oracle.h - TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's usearch_global hot path (+ usearch_local, the gapped x-drop, the pair
 * filters), used only as
 * the parity checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg.  Nothing under usearch12_amd/ may include, link or call this.
 *
 * Pinning: validated bit-exact against the compiled, unmodified reference
 * (oracle/_ref/usearch12, built by oracle/build_ref.sh) on the golden fixtures under
 * tests/golden/ (generators tests/golden/make_golden*.py; tests/test_oracle_*.py, tests/test_udb.py).
 */
#ifndef UGS_ORACLE_H
#define UGS_ORACLE_H

#include <stdint.h>
#include "../include/ugs.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_db orc_db;

int  orc_db_create(const ugs_params *p, const char *seqs, const uint64_t *offs,
                   uint32_t nseq, orc_db **out);
void orc_db_destroy(orc_db *db);
/* keys for the pair filters / -abskew (see include/ugs.h ugs_db_set_pair_keys); query keys are borrowed until the next search */
int  orc_db_set_pair_keys(orc_db *db, const uint32_t *label_key, const uint32_t *size);
void orc_set_query_pair_keys(orc_db *db, const uint32_t *label_key, const uint32_t *size);

/* masked DB letters (same offsets as the input) */
const char *orc_db_masked(const orc_db *db);
/* CSR view of the UDB index: row_off[slots+1], postings[row_off[slots]] */
uint64_t orc_db_slots(const orc_db *db);
const uint64_t *orc_db_row_off(const orc_db *db);
const uint32_t *orc_db_postings(const orc_db *db);

/* whole search, nthreads std worker threads over queries; same output contract as ugs_search_batch */
int orc_search_batch(orc_db *db, const char *qseqs, const uint64_t *qoffs, uint32_t nq,
                     ugs_hit *hits, uint64_t hits_cap, uint32_t *nhits_per_query,
                     uint32_t *cigar_pool, uint64_t cigar_cap, uint64_t *cigar_used,
                     int nthreads);

/* algorithmic work counters of the last orc_search_batch */
typedef struct orc_stats {
  uint64_t postings, query_letters, target_letters, pairs_aligned, dp_cells, hits,
           ungapped_calls, dp_calls;
} orc_stats;
void orc_get_stats(const orc_db *db, orc_stats *st);

/* stage: ranked candidate list of one query strand (already reverse-complemented by the
 * caller if needed), in the order the candidate loop walks it; returns total kept count,
 * writes min(cap, kept) entries */
int orc_rank(orc_db *db, const char *q, uint32_t ql, uint32_t *cand, uint32_t *cnt, uint32_t cap);

/* stage: GlobalAligner::Align on one pair.  Returns 1 if aligned (path written as
 * NUL-terminated M/D/I text into path[cap]), 0 if rejected. hsp_fract_id receives HSPFractId. */
int orc_align_pair(orc_db *db, const char *q, uint32_t ql, const char *t, uint32_t tl,
                   char *path, uint32_t cap, float *hsp_fract_id);

/* stage: ViterbiFastMainDiagMem on a hole with explicit 12 gap penalties
 * pen = {OpenA,OpenB,ExtA,ExtB,LOpenA,LOpenB,LExtA,LExtB,ROpenA,ROpenB,RExtA,RExtB} */
float orc_viterbi_band(orc_db *db, const char *a, uint32_t la, const char *b, uint32_t lb,
                       uint32_t band, const float *pen, char *path, uint32_t cap, uint64_t *cells);

/* stage: FastMaskSeq in place */
void orc_fastmask(char *seq, uint32_t len);

/* reverse complement (seqinfo.cpp:292-323) */
void orc_revcomp(const char *seq, uint32_t len, char *out);

/* text writers (blast6out.cpp:27-80, outputuc.cpp:10-93) */
int orc_format_blast6(const ugs_hit *h, const char *qlabel, const char *tlabel, char *buf, int cap);
int orc_format_blast6_local(const ugs_params *p, const ugs_hit *h, const char *qlabel, const char *tlabel, char *buf, int cap);
int orc_format_uc_hit(const ugs_hit *h, const uint32_t *cigar_pool, int is_nucleo,
                      const char *qlabel, const char *tlabel, char *buf, int cap);
int orc_format_uc_nohit(uint32_t ql, const char *qlabel, char *buf, int cap);

/* gapped x-drop extension, SURVEY.md 8a X1-X3 (xdropfwdmem.cpp, xdropbwdmem.cpp, xdropfwdsplit.cpp,
 * xdropbwdsplit.cpp, xdropalignmem.cpp).  One job of the batched ABI (include/ugs.h ugs_xdrop_batch);
 * path receives the NUL-terminated M/D/I text (needs la+lb+2 bytes); hsp->path_* are left 0.
 * Returns 0, or -1 when the job violates the reference's asserts. */
void orc_xdrop_params_init(ugs_xdrop_params *p, int is_nucleo);
/* test statistic: times the reference's unguarded insert-branch re-initialisation (xdropfwdmem.cpp:625-631)
 * overwrote a freshly stored cell (not thread-safe; tests only) */
unsigned long orc_xdrop_wipes(void);
int orc_xdrop_job(const ugs_xdrop_params *p, const char *a, uint32_t la, const char *b, uint32_t lb,
                  const ugs_xdrop_job *job, ugs_xdrop_hsp *hsp, char *path, uint64_t *cells);

/* cluster_fast (clusterfast.cpp:81-133): see ugs_oracle.c for the output contract */
uint32_t orc_derep_full(const char *seqs, const uint64_t *offs, uint32_t nseq, int revcomp, uint32_t *seq_unique, uint32_t *uniq_seed);
void orc_order_desc_u32(const uint32_t *values, uint32_t n, uint32_t *order);
int orc_cluster_fast(const ugs_params *p, const char *seqs, const uint64_t *offs, uint32_t nseq,
                     uint32_t *seq_unique, uint32_t *uniq_seed, uint32_t *n_unique,
                     uint32_t *uniq_cluster, uint32_t *uniq_nhits, uint32_t *centroid_uniq, uint32_t *cluster_size,
                     uint32_t *n_clusters, ugs_hit *hits, uint64_t hits_cap, uint32_t *cigar_pool, uint64_t cigar_cap,
                     uint64_t *n_hits, uint64_t *cigar_used);
/* + -sort (0 unset, 1 length, 2 size), -sizein; size_in[nseq] = ;size= of every label, UINT32_MAX = none (may be NULL) */
int orc_cluster_fast_sorted(const ugs_params *p, const char *seqs, const uint64_t *offs, uint32_t nseq,
                     int sort_mode, const uint32_t *size_in, int sizein,
                     uint32_t *seq_unique, uint32_t *uniq_seed, uint32_t *n_unique,
                     uint32_t *uniq_cluster, uint32_t *uniq_nhits, uint32_t *centroid_uniq, uint32_t *cluster_size,
                     uint32_t *n_clusters, ugs_hit *hits, uint64_t hits_cap, uint32_t *cigar_pool, uint64_t cigar_cap,
                     uint64_t *n_hits, uint64_t *cigar_used);

void orc_params_init(ugs_params *p, int is_nucleo, double id);
/* switch a parameter block to usearch_local (searcher.cpp:28-50, localmulti.cpp, localaligner.cpp, estats.cpp);
 * id_set = 0 drops the identity filter and ranks with the 0.5 fallback */
void orc_params_set_local(ugs_params *p, double evalue, int id_set);
/* Karlin-Altschul numbers of a local hit (estats.cpp:72-96): E = QL * DBSize / 2^bits */
void orc_local_evalue(const ugs_params *p, double raw, uint32_t ql, double *evalue, double *bits);
/* test statistic: local hits whose rescored path (AlignResult::GetRawScore) differs from the x-drop score */
unsigned long orc_local_rescore_diffs(void);

#ifdef __cplusplus
}
#endif
#endif
