// ugs_host.cpp - the C-ABI (include/ugs.h) over the HIP kernels: handles, launch geometry,
// HBM residency, hit gathering, and the blast6/uc text writers.  No CPU compute fallback:
// every search entry point needs a gfx950 device.
#include "ugs_host.h"
#include <atomic>

#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

static thread_local char g_err[512] = "";
void ugs_set_error(const char *fmt, ...)
{
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char *ugs_last_error(void) { return g_err; }

extern "C" int ugs_abi_version(void) { return UGS_ABI_VERSION; }

extern "C" int ugs_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int ugs_device_synchronize(int device)
{
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipDeviceSynchronize());
  return UGS_OK;
}

// o_defaults.inc, terminator.cpp:26-31, udbparams.cpp:235-261, alnheuristics.cpp:36,44;
// option values are stored as float in the reference (opts.cpp:265)
extern "C" int ugs_params_init(ugs_params *p, int is_nucleo, double id)
{
  if (!p) return UGS_E_ARG;
  memset(p, 0, sizeof(*p));
  p->is_nucleo = is_nucleo ? 1 : 0;
  p->word_len = is_nucleo ? 8 : 5;
  p->id = (float)id;
  p->id_accept = (double)(float)id;
  p->id_set = 1;
  p->strand_both = 0;
  p->max_accepts = 1; p->max_rejects = 32;
  p->big = 100000; p->bump_pct = 50; p->stepwords = 8;
  p->band = 16; p->minhsp = 16; p->xdrop_nw = 8.0f;
  p->match = 1.0f; p->mismatch = -2.0f;
  p->hsp_word_len = is_nucleo ? 5 : 3;
  p->dbmask = 1;
  p->xdrop_u = 16.0f; p->xdrop_g = 32.0f;         // o_defaults.inc:20,22
  p->local_open = -10.0f; p->local_ext = -1.0f;   // alnparams.cpp:362-369 (the -lopen/-lext defaults count as set)
  p->ka_dbsize = 1e9f;                            // o_defaults.inc:2
  p->max_hsps = 8;
  return UGS_OK;
}

// usearch_local: -evalue is required; without -id the accepter has no identity test and the ranking uses 0.5
// (makedbsearcher.cpp:172 oget_fltd(OPT_id, 0.5))
extern "C" int ugs_params_set_local(ugs_params *p, double evalue, int id_set)
{
  if (!p || !(evalue > 0.0)) { ugs_set_error("usearch_local needs -evalue > 0"); return UGS_E_ARG; }
  p->local = 1;
  p->evalue = (float)evalue;
  if (!id_set) { p->id = 0.5f; p->id_accept = 0.5; p->id_set = 0; }
  return UGS_OK;
}

// Karlin-Altschul statistics, estats.cpp:25-99.  The reference is built with -O3 -ffast-math (its Makefile:11-14)
// and what it prints depends on that: gcc folds (x/Log2)*Log2, multiplies by 1/ln 2 instead of dividing and turns
// NM/pow(2,bits) into exp2(-bits)*DBSize*QL (which keeps e-values below 1e-292 representable).  These are the
// operations of the compiled code (estats.o, gcc 11.4).
namespace {
struct EStats {
  double GappedLambda, UngappedLambda, LogGappedK, LogUngappedK, DBSize, MaxEvalue;
  explicit EStats(const ugs_params &p)
  {
    if (p.is_nucleo) { GappedLambda = 1.280; UngappedLambda = 1.330; LogGappedK = log(0.460); LogUngappedK = log(0.621); }
    else { GappedLambda = 0.267; UngappedLambda = 0.311; LogGappedK = log(0.0410); LogUngappedK = log(0.128); }
    DBSize = (double)p.ka_dbsize;                 // makedbsearcher.cpp:89-95: a float
    MaxEvalue = (double)p.evalue;
  }
  double min_ungapped(uint32_t QL) const { return ((log((double)QL * DBSize) + LogUngappedK) - log(MaxEvalue)) / UngappedLambda; }
  double bits(double raw) const { return (raw * GappedLambda - LogGappedK) * 1.4426950408889634; }
  double evalue(double raw, uint32_t QL) const { return (exp2(-bits(raw)) * DBSize) * (double)QL; }
};
}  // namespace

extern "C" int ugs_local_evalue(const ugs_params *p, double raw_score, uint32_t ql, double *evalue, double *bits)
{
  if (!p) return UGS_E_ARG;
  const EStats es(*p);
  if (bits) *bits = es.bits(raw_score);
  if (evalue) *evalue = es.evalue(raw_score, ql);
  return UGS_OK;
}

// ---------------------------------------------------------------- tables
extern const char UGS_B62_ORDER[] = "ARNDCQEGHILKMFPSTWYVBZX";
#define B62_ORDER UGS_B62_ORDER
extern const signed char UGS_B62[23][23] = {   // BLOSUM62 (NCBI), the 23 alphabetic symbols
  { 4,-1,-2,-2, 0,-1,-1, 0,-2,-1,-1,-1,-1,-2,-1, 1, 0,-3,-2, 0,-2,-1, 0},
  {-1, 5, 0,-2,-3, 1, 0,-2, 0,-3,-2, 2,-1,-3,-2,-1,-1,-3,-2,-3,-1, 0,-1},
  {-2, 0, 6, 1,-3, 0, 0, 0, 1,-3,-3, 0,-2,-3,-2, 1, 0,-4,-2,-3, 3, 0,-1},
  {-2,-2, 1, 6,-3, 0, 2,-1,-1,-3,-4,-1,-3,-3,-1, 0,-1,-4,-3,-3, 4, 1,-1},
  { 0,-3,-3,-3, 9,-3,-4,-3,-3,-1,-1,-3,-1,-2,-3,-1,-1,-2,-2,-1,-3,-3,-2},
  {-1, 1, 0, 0,-3, 5, 2,-2, 0,-3,-2, 1, 0,-3,-1, 0,-1,-2,-1,-2, 0, 3,-1},
  {-1, 0, 0, 2,-4, 2, 5,-2, 0,-3,-3, 1,-2,-3,-1, 0,-1,-3,-2,-2, 1, 4,-1},
  { 0,-2, 0,-1,-3,-2,-2, 6,-2,-4,-4,-2,-3,-3,-2, 0,-2,-2,-3,-3,-1,-2,-1},
  {-2, 0, 1,-1,-3, 0, 0,-2, 8,-3,-3,-1,-2,-1,-2,-1,-2,-2, 2,-3, 0, 0,-1},
  {-1,-3,-3,-3,-1,-3,-3,-4,-3, 4, 2,-3, 1, 0,-3,-2,-1,-3,-1, 3,-3,-3,-1},
  {-1,-2,-3,-4,-1,-2,-3,-4,-3, 2, 4,-2, 2, 0,-3,-2,-1,-2,-1, 1,-4,-3,-1},
  {-1, 2, 0,-1,-3, 1, 1,-2,-1,-3,-2, 5,-1,-3,-1, 0,-1,-3,-2,-2, 0, 1,-1},
  {-1,-1,-2,-3,-1, 0,-2,-3,-2, 1, 2,-1, 5, 0,-2,-1,-1,-1,-1, 1,-3,-1,-1},
  {-2,-3,-3,-3,-2,-3,-3,-3,-1, 0, 0,-3, 0, 6,-4,-2,-2, 1, 3,-1,-3,-3,-1},
  {-1,-2,-2,-1,-3,-1,-1,-2,-2,-3,-3,-1,-2,-4, 7,-1,-1,-4,-3,-2,-2,-1,-2},
  { 1,-1, 1, 0,-1, 0, 0, 0,-1,-2,-2, 0,-1,-2,-1, 4, 1,-3,-2,-2, 0, 0, 0},
  { 0,-1, 0,-1,-1,-1,-1,-2,-2,-1,-1,-1,-1,-2,-1, 1, 5,-2,-2, 0,-1,-1, 0},
  {-3,-3,-4,-4,-2,-2,-3,-2,-2,-3,-2,-3,-1, 1,-4,-3,-2,11, 2,-3,-4,-3,-2},
  {-2,-2,-2,-3,-2,-1,-2,-3, 2,-1,-1,-2,-1, 3,-3,-2,-2, 2, 7,-1,-3,-2,-1},
  { 0,-3,-3,-3,-1,-2,-2,-3,-3, 3, 1,-2, 1,-1,-2,-2, 0,-3,-1, 4,-3,-2,-1},
  {-2,-1, 3, 4,-3, 0, 1,-1, 0,-3,-4, 0,-3,-3,-2, 0,-1,-4,-3,-3, 4, 1,-1},
  {-1, 0, 0, 1,-3, 3, 4,-2, 0,-3,-3, 1,-1,-3,-1, 0,-1,-3,-2,-2, 1, 4,-1},
  { 0,-1,-1,-1,-2,-1,-1,-1,-1,-1,-1,-1,-1,-1,-2, 0, 0,-2,-1,-1,-1,-1,-1},
};

static int build_tables(const ugs_params &p, UgsTables &T)
{
  memset(&T, 0, sizeof(T));
  const bool nt = p.is_nucleo != 0;
  int let[26];                               // alphabet letter per 'A'+k, -1 = not in the word alphabet
  for (int k = 0; k < 26; ++k) let[k] = -1;
  if (nt) { let['A' - 'A'] = 0; let['C' - 'A'] = 1; let['G' - 'A'] = 2; let['T' - 'A'] = 3; let['U' - 'A'] = 3; }
  else { const char *aa = "ACDEFGHIKLMNPQRSTVWY"; for (int i = 0; i < 20; ++i) let[aa[i] - 'A'] = i; }
  for (int c = 0; c < 256; ++c) {
    T.udb_letter[c] = 0xff; T.hsp_letter[c] = 0; T.cls[c] = 31; T.comp[c] = (uint8_t)c;
    const bool up = c >= 'A' && c <= 'Z', lo = c >= 'a' && c <= 'z';
    if (up || lo) {
      const int k = up ? c - 'A' : c - 'a';
      T.cls[c] = (uint8_t)(k | (lo ? 32 : 0));
      if (let[k] >= 0) { T.hsp_letter[c] = (uint8_t)let[k]; if (up) T.udb_letter[c] = (uint8_t)let[k]; }
    }
  }
  { // reverse complement map; lower-case 'u' is absent from the reference table (alpha.cpp:3005-3265)
    const char *from = "ABCDGHKMNRSTUVWXY", *to = "TVGHCDMKNYSAABWXR";
    for (int k = 0; from[k]; ++k) {
      T.comp[(unsigned char)from[k]] = (uint8_t)to[k];
      if (from[k] != 'U') T.comp[(unsigned char)(from[k] | 0x20)] = (uint8_t)(to[k] | 0x20);
    }
  }
  // 2 x substitution score by letter
  if (nt) {
    const float m2 = p.match * 2.0f, mm2 = p.mismatch * 2.0f;
    if (m2 != floorf(m2) || mm2 != floorf(mm2) || fabsf(m2) > 120 || fabsf(mm2) > 120) {
      ugs_set_error("match/mismatch must be multiples of 0.5 within +-60 for the integer DP"); return UGS_E_ENVELOPE;
    }
    for (int i = 0; i < 26; ++i) for (int j = 0; j < 26; ++j)
      if (let[i] >= 0 && let[j] >= 0) T.sub2[i * 32 + j] = (int8_t)(let[i] == let[j] ? m2 : mm2);
  } else {
    for (int i = 0; i < 23; ++i) for (int j = 0; j < 23; ++j)
      T.sub2[(B62_ORDER[i] - 'A') * 32 + (B62_ORDER[j] - 'A')] = (int8_t)(2 * UGS_B62[i][j]);
  }
  // identity classes (alpha2.cpp:220-300): same letter; nt: IUPAC "base in wildcard set" either way;
  // aa: X matches anything, and UPPER-CASE B/N, B/D, Z/Q, Z/E
  int nucbit[26], iupac[26];
  for (int k = 0; k < 26; ++k) nucbit[k] = iupac[k] = 0;
  nucbit['A' - 'A'] = 1; nucbit['C' - 'A'] = 2; nucbit['G' - 'A'] = 4; nucbit['T' - 'A'] = 8; nucbit['U' - 'A'] = 8;
  for (int k = 0; k < 26; ++k) iupac[k] = nucbit[k];
  { const char *codes = "MRWSYKVHDBXN"; const char *sets[] = {"AC", "AG", "AT", "CG", "CT", "GT", "ACG", "ACT", "AGT", "CGT", "GATC", "GATC"};
    for (int q = 0; codes[q]; ++q) { int b = 0; for (const char *s = sets[q]; *s; ++s) b |= nucbit[*s - 'A']; iupac[codes[q] - 'A'] = b; } }
  for (int a = 0; a < 64; ++a) for (int b = 0; b < 64; ++b) {
    const int ka = a & 31, kb = b & 31;
    if (ka >= 26 || kb >= 26) continue;
    bool m = (ka == kb);
    if (!m) {
      if (nt) m = (nucbit[ka] & iupac[kb]) || (nucbit[kb] & iupac[ka]);
      else {
        m = (ka == 'X' - 'A') || (kb == 'X' - 'A');
        if (!m && a < 32 && b < 32) {
          auto pr = [&](char x, char y) { return (ka == x - 'A' && kb == y - 'A') || (ka == y - 'A' && kb == x - 'A'); };
          m = pr('B', 'N') || pr('B', 'D') || pr('Z', 'Q') || pr('Z', 'E');
        }
      }
    }
    if (m) T.match[a] |= (1ull << b);
  }
  return UGS_OK;
}

// wordparams.cpp:60-112 (table from CD-HIT as the reference holds it)
static const double MinWordFractAmino[50] = {
  0.00, 0.00, 0.00, 0.00, 0.01, 0.01, 0.01, 0.02, 0.02, 0.02, 0.03, 0.04, 0.04, 0.05, 0.06, 0.06, 0.08,
  0.08, 0.10, 0.10, 0.11, 0.14, 0.14, 0.14, 0.17, 0.17, 0.18, 0.20, 0.21, 0.21, 0.27, 0.28, 0.31, 0.34,
  0.36, 0.41, 0.43, 0.45, 0.48, 0.54, 0.55, 0.56, 0.64, 0.69, 0.73, 0.75, 0.80, 0.85, 0.90, 0.95};

// wordparams.cpp:125-135,145-159,167-192: QueryStep for Nu unique query words.  Evaluated on the
// host in fp64 exactly as the reference does (no FMA contraction) and uploaded as a table.
static uint32_t query_step(const ugs_params &p, uint32_t Nu)
{
  volatile double FractId = (double)(p.id_set ? p.id : 0.5f);     // makedbsearcher.cpp:195
  uint32_t Thresh;
  if (p.is_nucleo) {
    volatile double d = 1 - FractId;
    volatile double e = d * p.word_len;
    volatile double WordFract = 1 - e;
    if (WordFract < 0.0) Thresh = 1;
    else {
      volatile double wf = WordFract * Nu;
      Thresh = wf < 1.0 ? 1u : (uint32_t)wf;
    }
  } else {
    if (FractId < 0.5) Thresh = 0;
    else {
      volatile double x = (FractId - 0.5) * 100;
      uint32_t i = (uint32_t)x;
      if (i >= 50) i = 49;
      volatile double y = MinWordFractAmino[i] * Nu;
      Thresh = (uint32_t)y;
    }
  }
  if (p.stepwords == 0) return 1;
  uint32_t Step = Thresh / p.stepwords;
  return Step == 0 ? 1 : Step;
}

// ---------------------------------------------------------------- db
extern "C" void ugs_db_destroy(ugs_db *db)
{
  if (!db) return;
  (void)hipSetDevice(db->device);
  // no stream or event is destroyed while anything of this device is in flight (a completion signal is a 64-bit word the runtime
  // DECREMENTS: DESIGN section 4, "host heap").  The barrier is DEVICE-WIDE on purpose and stalls other handles / libraries on the same
  // GPU for the moment of a destroy: r5 tried the handle's own streams only and one of two full GPU suites then died of a
  // SIGABRT inside a later ugs_db_create - the r02 symptom - while every suite with the device-wide barrier has passed.
  if (db->stream) (void)hipStreamSynchronize(db->stream);
  (void)hipDeviceSynchronize();
  (void)ugs_free(db->d_pk);
  (void)ugs_free(db->d_seqs); (void)ugs_free(db->d_offs); (void)ugs_free(db->d_row_off); (void)ugs_free(db->d_postings); (void)ugs_free(db->d_part); (void)ugs_free(db->d_part2);
  (void)ugs_free(db->d_post16);
  (void)ugs_free(db->d_row_off2); (void)ugs_free(db->d_postings2);
  (void)ugs_free(db->d_step); (void)ugs_free(db->d_tab); (void)ugs_free(db->d_xsub2); (void)ugs_free(db->d_xcls); (void)ugs_free(db->d_tkey); (void)ugs_free(db->d_tsize);
  if (db->stream) (void)hipStreamDestroy(db->stream);
  if (db->setup_stream) (void)hipStreamDestroy(db->setup_stream);
  delete db;
}

static int db_step_table(ugs_db *db, uint32_t n)
{
  if (db->step.size() >= n) return UGS_OK;
  size_t old = db->step.size();
  db->step.resize(n);
  for (size_t i = old; i < n; ++i) db->step[i] = query_step(db->p, (uint32_t)i);
  if (db->d_step) HIPCHK(ugs_free(db->d_step));
  HIPCHK(ugs_malloc(&db->d_step, n * sizeof(uint32_t)));
  HIPCHK(ugs_h2d(db->d_step, db->step.data(), n * sizeof(uint32_t), db->stream));
  HIPCHK(hipStreamSynchronize(db->stream));
  db->v.step_tab = db->d_step; db->v.step_n = (uint32_t)n;
  return UGS_OK;
}

static int env_int(const char *name, int lo, int hi, int unset)
{
  const char *e = getenv(name);
  if (!e || !*e) return unset;
  const long v = atol(e);
  return (v >= lo && v <= hi) ? (int)v : unset;
}

// the debug switches (ugs_host.h UgsTune; listed in include/ugs.h), read once per database handle
UgsTune ugs_tune_read()
{
  UgsTune t;
  t.no_packed = getenv("UGS_NO_PACKED") != nullptr;
  t.longrows = env_int("UGS_LONGROWS", 0, 1, -1);
  t.gsize = env_int("UGS_GSIZE", 64, 65536, 0); if (t.gsize % 64) t.gsize = 0;
  t.gshift = env_int("UGS_GSHIFT", 6, 16, 0);
  t.rank_wgs = env_int("UGS_RANK_WGS_PER_CU", 1, 8, 0);
  t.align_wgs = env_int("UGS_ALIGN_WGS_PER_CU", 1, 8, 0);
  { const char *e = getenv("UGS_EMIT_LIMIT"); const long v = e ? atol(e) : 0; t.emit_limit = v >= 1 ? v : 0; }
  t.debug_sync = getenv("UGS_DEBUG_SYNC") != nullptr;
  t.phase_clocks = getenv("UGS_PHASE_CLOCKS") != nullptr;
  t.wide_offsets = getenv("UGS_WIDE_OFFSETS") != nullptr;
  t.rank2 = env_int("UGS_RANK2", 0, 1, -1);
  t.r2_g = env_int("UGS_R2_G", 8192, 65536, 0); if (t.r2_g % 8192) t.r2_g = 0;
  t.r2_kcap = env_int("UGS_R2_KCAP", 8, 4096, 0);
  t.r2_waves = env_int("UGS_R2_WAVES", 1, 32, 0);
  t.r2_clcap = env_int("UGS_R2_CLCAP", 48, 4096, 0);
  t.r2_p16 = env_int("UGS_R2_P16", 0, 1, -1);
  t.r2_hv = env_int("UGS_R2_HV", 0, 2, -1);
  t.setup_stream = env_int("UGS_SETUP_STREAM", 0, 1, 0);
  t.r3 = env_int("UGS_R3", 0, 1, -1);
  t.r3_sp = env_int("UGS_R3_SP", 1, 63, 0);
  t.r3_pps = env_int("UGS_R3_PPS", 64, 1 << 20, 0);
  t.qpk = getenv("UGS_QPK") != nullptr;
  t.align_group = env_int("UGS_ALIGN_GROUP", 0, 64, -1);
  return t;
}

// Partition size and table, the Big latch and the index-dependent fields of the device view; called when the index was
// built (ugs_db_create) or grew (ugs_db_append).
int ugs_db_replan(ugs_db *db)
{
  const uint32_t nseq = db->v.nseq, slots = db->v.slots;
  // ~40 postings per (row, partition): a sub-row then (almost) never exceeds one wavefront
  // (Poisson tail < 1e-3), while 60 % of the lanes of every row instruction carry a posting
  uint32_t gsize = 8192;
  {
    double avg_row = db->n_postings ? (double)db->n_postings / (double)slots : 1.0;
    double g = 40.0 * (double)(nseq ? nseq : 1) / (avg_row > 1.0 ? avg_row : 1.0);
    uint64_t gs = ((uint64_t)g + 511) / 1024 * 1024;
    if (gs < 1024) gs = 1024;
    uint64_t budget = std::max<uint64_t>(64ull << 20, db->n_postings);
    db->sparse = false;
    if (gs > 65536) {
      // sparse rows (protein dictionaries: a few hundred postings per row): no partition size gives 40 postings per
      // sub-row; the scan then flattens all sub-rows of a range into one instruction stream and what matters is the
      // number of resident waves, i.e. a small counter table (16 Ki targets x 8 bits = 16 KiB of LDS per wave).
      // The partition table may grow up to the size of the postings themselves for it.
      gs = 16384;
      db->sparse = true;
      budget = std::max<uint64_t>(256ull << 20, db->n_postings * 4);
    }
    while (gs < 65536 && (uint64_t)slots * ((uint64_t)nseq / gs + 2) * 4 > budget) gs += 1024;
    gsize = (uint32_t)gs;
    // (queries with more than 255 words need 16-bit counters on the small path: partitions are then kept small enough for
    // four workgroups per CU, and their short sub-rows take the flattened scan - 2.4x on cluster_fast's small-path phase)
    if (db->gsize_limit && nseq <= db->p.big && gsize > db->gsize_limit) gsize = db->gsize_limit;
    if (db->tune.gsize) gsize = (uint32_t)db->tune.gsize;
    if (db->tune.gshift) gsize = 1u << db->tune.gshift;
  }
  const uint32_t np = nseq ? (uint32_t)(((uint64_t)nseq - 1) / gsize + 1) : 1;
  const uint64_t need = (uint64_t)slots * (np + 1);
  if (!db->d_part || need > db->part_cap) {
    if (db->d_part) HIPCHK(ugs_free(db->d_part));
    db->d_part = nullptr;
    db->part_cap = need + need / 4;
    HIPCHK(ugs_malloc(&db->d_part, (size_t)db->part_cap * sizeof(uint32_t)));
  }
  RCCHK(ugs_build_part(db->d_row_off, db->d_postings, slots, np, gsize, db->d_part, db->stream));
  HIPCHK(hipStreamSynchronize(db->stream));
  // The bitmap ranking kernel (ugs_rank2.hip) for dense Big-path indexes: one LDS bit per target, so its partitions are
  // larger (G2 targets, a multiple of 8192 up to 65536) and have a table of their own.  G2 is chosen so that a (row,
  // partition) sub-row averages <= ~205 postings (7 of 8 lanes of the 256-posting load busy, a second chunk rare): it is read as ONE 16-byte load per lane (256 postings per wave
  // instruction).  Keys carry 24-bit targets, hence nseq <= 2^24; sparse dictionaries (protein) keep k_rank.
  uint32_t gsize2 = 0, np2 = 0;
  if (nseq > db->p.big && !db->sparse && nseq <= (1u << 24) && db->n_postings && db->tune.rank2 != 0) {
    const double dens = ((double)db->n_postings / (double)slots) / (double)nseq;        // a row's postings per target
    uint64_t g = dens > 0 ? (uint64_t)(205.0 / dens) / 8192 * 8192 : 65536;
    g = std::min<uint64_t>(65536, std::max<uint64_t>(8192, g));
    if (db->tune.r2_g) g = (uint64_t)db->tune.r2_g;
    if (dens * (double)g >= 64.0 || db->tune.rank2 == 1) { gsize2 = (uint32_t)g; np2 = (uint32_t)(((uint64_t)nseq - 1) / gsize2 + 1); }
  }
  // sparse Big-path index (protein): the gather variant (k_rank2g) - one chunk holds the sub-rows of all sampled rows of a partition, so
  // the partitions are as large as the bitmap allows; its row lanes read a table line of at most 64 words and address the postings with
  // 32-bit byte offsets
  db->r2_gather = false;
  if (nseq > db->p.big && db->sparse && nseq <= (1u << 24) && db->n_postings && db->n_postings < (1ull << 30) - 1024 && db->tune.rank2 != 0) {
    const uint64_t g = db->tune.r2_g ? (uint64_t)db->tune.r2_g : 65536;
    const uint64_t n2 = ((uint64_t)nseq - 1) / g + 1;
    if (n2 <= 63 && g % 8192 == 0 && g <= 65536) { gsize2 = (uint32_t)g; np2 = (uint32_t)n2; db->r2_gather = true; }
  }
  if (gsize2) {
    const uint64_t need2 = (uint64_t)slots * (np2 + 1);
    if (!db->d_part2 || need2 > db->part2_cap) {
      if (db->d_part2) HIPCHK(ugs_free(db->d_part2));
      db->d_part2 = nullptr;
      db->part2_cap = need2 + need2 / 4;
      HIPCHK(ugs_malloc(&db->d_part2, (size_t)db->part2_cap * sizeof(uint32_t)));
    }
    RCCHK(ugs_build_part(db->d_row_off, db->d_postings, slots, np2, gsize2, db->d_part2, db->stream));
    HIPCHK(hipStreamSynchronize(db->stream));
  }
  ++db->index_gen;                      // (d_post16, if any, belongs to the index before this call)
  UgsDbView &v = db->v;
  v.seqs = db->d_seqs; v.offs = db->d_offs; v.row_off = db->d_row_off; v.postings = db->d_postings; v.part = db->d_part;
  v.part2 = gsize2 ? db->d_part2 : nullptr; v.np2 = np2; v.gsize2 = gsize2;
  v.group_after = db->tune.align_group < 0 ? 1u : (uint32_t)db->tune.align_group;
  v.pk = db->tune.no_packed ? nullptr : db->d_pk;       // UGS_NO_PACKED: k_align fetches every target from the byte array (A/B, fault isolation)
  v.np = np; v.gsize = gsize; v.big = nseq > db->p.big ? 1 : 0; v.max_tlen = db->max_tlen;
  db->hbm_bytes = db->nletters + ((size_t)nseq + 1) * 8 + ((size_t)slots + 1) * 8 + db->n_postings * 4 +
                  (size_t)slots * (np + 1) * 4 + (gsize2 ? (size_t)slots * (np2 + 1) * 4 : 0) + sizeof(UgsTables);
  return UGS_OK;
}

extern "C" int ugs_db_create(const ugs_params *p, const char *seqs, const uint64_t *offs, uint32_t nseq,
                             int device, ugs_db **out)
{
  if (!p || !offs || !out || (nseq && !seqs)) { ugs_set_error("null argument"); return UGS_E_ARG; }
  int ndev = ugs_device_count();
  if (device < 0 || device >= ndev) {
    ugs_set_error("device %d not available (%d devices); there is no CPU fallback", device, ndev);
    return UGS_E_NODEVICE;
  }
  // 0 = unlimited (terminator.cpp:40-45,91-97 test "m_MaxAccepts > 0 && ..."): the walk then ends only on the other limit
  // or at the end of the candidate list; the device keeps UGS_KMAX candidates per strand and a walk that would need more
  // fails loudly at ugs_batch_sync (never a silently shortened walk)
  const bool open_walk = p->max_accepts == 0 || p->max_rejects == 0;
  if (p->max_accepts < 0 || p->max_rejects < 0) { ugs_set_error("max_accepts/max_rejects must be >= 0 (0 = unlimited)"); return UGS_E_ARG; }
  // a walk that may visit more than the UGS_KMAX candidates a ranking pass keeps (-maxrejects 128, -maxaccepts 0 ...): the search runs
  // as usual and the walks that used up their list are continued over the unit's complete sorted list (deep walks, ugs_deep.hip)
  // ... and the small path with -selfid: pairs that filter passes over are not counted (searcher.cpp:63-67), so a walk among many identical
  // sequences may want any number of candidates (r4: UGS_ERR_PAIRCAP beyond 32 spare ones)
  const bool termid_on = (p->align_flags & (UGS_A_TERMID | UGS_A_TERMIDD)) != 0;
  const bool deep_by_depth = open_walk || (int64_t)p->max_accepts + p->max_rejects - 1 > UGS_KMAX;
  // (-selfid together with -termid / -termidd keeps round 4's scheme - K + 32 spare candidates, UGS_ERR_PAIRCAP if a walk wants more -
  //  instead of being refused as "deeper than 64": its configured depth is not)
  const bool deep_walk = deep_by_depth || (!termid_on && (p->pair_mask & UGS_P_SELFID) && !p->local && (uint64_t)nseq <= (uint64_t)p->big);
  if (deep_by_depth && termid_on) {
    // (-termid / -termidd look at the hits of both strands of a query in walk order: a parked plus-strand walk would have to finish first)
    ugs_set_error("-termid / -termidd with more than %d candidates per walk (max_accepts + max_rejects - 1, or an unlimited walk) are outside the device envelope", UGS_KMAX); return UGS_E_ENVELOPE;
  }
  const int alpha = p->is_nucleo ? 4 : 20;
  uint64_t slots64 = 1;
  for (int i = 0; i < p->word_len; ++i) { slots64 *= alpha; if (slots64 > (1ull << 28)) break; }
  uint64_t hspw64 = 1;
  for (int i = 0; i < p->hsp_word_len; ++i) { hspw64 *= alpha; if (hspw64 > 65536) break; }
  if (p->word_len < 1 || slots64 > (1ull << 28) || p->hsp_word_len < 1 || hspw64 > 65536 || p->band < 0) {
    ugs_set_error("unsupported word_len/hsp_word_len/band"); return UGS_E_ENVELOPE;
  }
  if (p->strand_both && !p->is_nucleo) { ugs_set_error("strand_both needs a nucleotide search"); return UGS_E_ARG; }
  if (p->local && p->align_flags) { ugs_set_error("-fulldp / -gaforce belong to the global aligner"); return UGS_E_ARG; }
  if (p->local && (p->pair_mask || (p->filter_mask & UGS_F_ABSKEW))) {
    // (the reference's small-path local walk aligns against the PREVIOUS target when SetTarget refuses a pair)
    ugs_set_error("pair filters / -abskew are implemented for usearch_global only"); return UGS_E_ARG;
  }
  if (p->local) {
    const float o2 = p->local_open * 2.0f, e2 = p->local_ext * 2.0f;
    if (!(p->evalue > 0.0f) || !(p->ka_dbsize > 0.0f) || p->max_hsps < 1 || !(p->xdrop_u >= 0.0f) || !(p->xdrop_g >= 0.0f)) {
      ugs_set_error("usearch_local needs evalue > 0, ka_dbsize > 0, max_hsps >= 1, xdrop_u/xdrop_g >= 0"); return UGS_E_ARG;
    }
    if (o2 != floorf(o2) || e2 != floorf(e2) || !(o2 < 0) || !(e2 < 0) || o2 < -2000 || e2 < -2000) {
      ugs_set_error("local_open/local_ext must be negative multiples of 0.5"); return UGS_E_ENVELOPE;
    }
    if ((uint64_t)p->max_accepts * p->max_hsps > 4096) { ugs_set_error("max_accepts * max_hsps > 4096"); return UGS_E_ENVELOPE; }
  }
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  ugs_db *db = new ugs_db();
  memset(&db->v, 0, sizeof(db->v));
  db->p = *p; db->device = device; db->num_cu = prop.multiProcessorCount;
  db->tune = ugs_tune_read();
  db->d_part2 = nullptr; db->part2_cap = 0;
  db->d_post16 = nullptr; db->post16_cap = 0; db->index_gen = 0; db->post16_gen = 0;
  db->d_seqs = nullptr; db->d_offs = nullptr; db->d_row_off = nullptr; db->d_postings = nullptr; db->d_part = nullptr;
  db->d_pk = nullptr; db->pack_cap = 0;
  db->d_row_off2 = nullptr; db->d_postings2 = nullptr; db->post_cap2 = 0; db->gsize_limit = 0;
  db->d_step = nullptr; db->d_tab = nullptr; db->stream = nullptr; db->setup_stream = nullptr; db->d_xsub2 = nullptr; db->d_xcls = nullptr; db->d_tkey = nullptr; db->d_tsize = nullptr; db->have_tkey = db->have_tsize = false; db->sparse = false;
  memset(&db->lv, 0, sizeof(db->lv));
  int rc = UGS_OK;
  auto fail = [&](int code) { ugs_db_destroy(db); return code; };
  if (hipStreamCreate(&db->stream) != hipSuccess) { ugs_set_error("hipStreamCreate failed"); return fail(UGS_E_HIP); }
  if (db->tune.setup_stream && hipStreamCreateWithFlags(&db->setup_stream, hipStreamNonBlocking) != hipSuccess) { ugs_set_error("hipStreamCreate failed"); return fail(UGS_E_HIP); }
  const uint64_t nletters = offs[nseq];
  uint32_t max_tlen = 0;
  for (uint32_t t = 0; t < nseq; ++t) {
    uint64_t L = offs[t + 1] - offs[t];
    if (L > 65535) { ugs_set_error("target %u longer than 65535 letters", t); return fail(UGS_E_ENVELOPE); }
    if (L > max_tlen) max_tlen = (uint32_t)L;
  }
  db->max_tlen = max_tlen;
  UgsTables T;
  if ((rc = build_tables(*p, T)) != UGS_OK) return fail(rc);
#define DBCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ugs_set_error("%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return fail(UGS_E_HIP); } } while (0)
  DBCHK(ugs_malloc(&db->d_tab, sizeof(UgsTables)));
  DBCHK(ugs_h2d(db->d_tab, &T, sizeof(T), db->stream));
  DBCHK(ugs_malloc(&db->d_seqs, nletters + 64));          // padded: the aligner prefetches letters as unaligned dwords
  DBCHK(ugs_malloc(&db->d_offs, ((size_t)nseq + 1) * sizeof(uint64_t)));
  if (nletters) DBCHK(ugs_h2d(db->d_seqs, seqs, nletters, db->stream));      // (through the library's page-locked staging buffer: ugs_host.h)
  DBCHK(ugs_h2d(db->d_offs, offs, ((size_t)nseq + 1) * sizeof(uint64_t), db->stream));
  if (p->dbmask < 0 || p->dbmask > 3) { ugs_set_error("dbmask must be 0..3"); return fail(UGS_E_ARG); }
  if ((rc = ugs_launch_mask(db->d_seqs, db->d_offs, nseq, p->dbmask == 3 ? (1 | ((p->is_nucleo ? 'N' : 'X') << 8)) : p->dbmask, db->stream)) != UGS_OK) return fail(rc);
  if (p->is_nucleo) {                                       // the masked letters packed 2 bits each (UgsDbView::pk)
    db->pack_cap = (nletters + 64) / 16 + 8;
    DBCHK(ugs_malloc(&db->d_pk, db->pack_cap * 8));
    DBCHK(hipMemsetAsync(db->d_pk, 0, db->pack_cap * 8, db->stream));
    if ((rc = ugs_launch_pack(db->d_tab, db->d_seqs, 0, (nletters + 15) / 16, db->d_pk, db->stream)) != UGS_OK) return fail(rc);
  }
  const uint32_t slots = (uint32_t)slots64;
  if ((rc = ugs_build_index(db->d_tab, db->d_seqs, db->d_offs, nseq, nletters, p->word_len, alpha, slots,
                            &db->d_row_off, &db->d_postings, &db->n_postings, &db->max_row, db->stream)) != UGS_OK)
    return fail(rc);
  db->nletters = nletters; db->seq_cap = nletters + 64; db->off_cap = (uint64_t)nseq + 1; db->post_cap = db->n_postings + 256; db->part_cap = 0;
  db->v.nseq = nseq; db->v.slots = slots;
  if ((rc = ugs_db_replan(db)) != UGS_OK) return fail(rc);
#undef DBCHK
  UgsDbView &v = db->v;
  v.tab = db->d_tab;
  v.word_len = p->word_len; v.alpha = alpha; v.bump_pct = p->bump_pct;
  v.hsp_w = p->hsp_word_len; v.hsp_words = (int)hspw64;
  v.xdrop2 = (int)floor(2.0 * (double)p->xdrop_nw);
  // alnheuristics.cpp:26-62 (float arithmetic as in the reference)
  const float idf = p->id_set ? p->id : 0.5f;
  float minfid, minscore;
  if (p->is_nucleo) { minfid = idf > 0.75f ? idf : 0.75f; minscore = minfid * (float)p->minhsp * p->match; }
  else {
    float mind = 9e9f;
    for (const char *a = "ACDEFGHIKLMNPQRSTVWY"; *a; ++a) { float s = (float)T.sub2[(*a - 'A') * 32 + (*a - 'A')] * 0.5f; if (s < mind) mind = s; }
    minfid = idf > 0.5f ? idf : 0.5f; minscore = minfid * mind * (float)p->minhsp;
  }
  v.min_hsp_fract_id = minfid;
  v.minscore2 = (int)ceil(2.0 * (double)minscore);
  v.min_hsp_len_opt = p->minhsp; v.band = p->band;
  v.open2 = p->is_nucleo ? -20 : -34; v.ext2 = -2; v.topen2 = -1; v.text2 = -1;   // alnparams.cpp:380-384
  v.id_accept = p->id_accept; v.id_set = p->id_set;
  v.filter_mask = p->filter_mask; v.maxid = p->maxid; v.query_cov = p->query_cov; v.max_query_cov = p->max_query_cov;
  v.target_cov = p->target_cov; v.max_target_cov = p->max_target_cov;
  v.mincols = p->mincols; v.maxgaps = p->maxgaps; v.maxdiffs = p->maxdiffs; v.mindiffs = p->mindiffs;
  v.max_accepts = p->max_accepts ? std::min(p->max_accepts, UGS_KMAX) : UGS_KMAX;      // hit slots per unit (deep walks chain blocks behind them)
  v.acc_limit = p->max_accepts ? p->max_accepts : 0x7fffffff;
  v.max_rejects = p->max_rejects ? p->max_rejects : 0x7fffffff;
  v.is_nucleo = p->is_nucleo; v.max_tlen = max_tlen;
  v.pair_mask = p->pair_mask; v.min_sizeratio = p->min_sizeratio; v.minqt = p->minqt; v.maxqt = p->maxqt; v.minsl = p->minsl; v.maxsl = p->maxsl;
  v.abskew = p->abskew; v.t_key = nullptr; v.t_size = nullptr;
  v.align_flags = p->align_flags | (open_walk ? UGS_A_OPENWALK : 0u) | (deep_walk ? UGS_A_DEEP : 0u); v.termid = p->termid; v.termidd = p->termidd;
  // every diagonal = ViterbiFastMem: -fulldp (globalalignmem.cpp:148-152) and -band 0 (:105-108,118-119: every hole, and a whole pair
  // without HSPs, goes through the unbanded aligner)
  if ((p->align_flags & UGS_A_FULLDP) || p->band == 0) v.band = 1 << 20;
  if ((rc = db_step_table(db, 4096)) != UGS_OK) return fail(rc);
  if (p->local) {   // x-drop tables (the ones ugs_xdrop_batch uses) and the constants of LocalAligner / XDropAlignMem
    int8_t xsub2[1024]; uint8_t xcls[256];
    ugs_xdrop_tables(p->is_nucleo, p->match * 2.0f, p->mismatch * 2.0f, xsub2, xcls);
    if (ugs_malloc(&db->d_xsub2, 1024) != hipSuccess || ugs_malloc(&db->d_xcls, 256) != hipSuccess ||
        ugs_h2d(db->d_xsub2, xsub2, 1024, db->stream) != hipSuccess ||
        ugs_h2d(db->d_xcls, xcls, 256, db->stream) != hipSuccess || hipStreamSynchronize(db->stream) != hipSuccess) { ugs_set_error("x-drop tables: HIP error"); return fail(UGS_E_HIP); }
    UgsLocalView &lv = db->lv;
    lv.sub2 = db->d_xsub2; lv.cls = db->d_xcls;
    lv.open2 = (int)(p->local_open * 2.0f); lv.ext2 = (int)(p->local_ext * 2.0f);
    lv.xdrop_g = p->xdrop_g; lv.abs_open = -p->local_open; lv.abs_ext = -p->local_ext; lv.xdrop_u = p->xdrop_u;
    lv.seed_w = (uint32_t)p->hsp_word_len;
    lv.W = std::min<uint32_t>(max_tlen, 4096) + 4;
  }
  *out = db;
  return UGS_OK;
}

// keys for the pair filters / -abskew (include/ugs.h)
static int upload_keys(uint32_t **d, const uint32_t *h, size_t n, bool *have, hipStream_t st)
{
  *have = false;
  if (!h) return UGS_OK;
  if (!*d) HIPCHK(ugs_malloc(d, std::max<size_t>(n, 1) * 4));
  if (n) HIPCHK(ugs_h2d(*d, h, n * 4, st));
  HIPCHK(hipStreamSynchronize(st));
  *have = true;
  return UGS_OK;
}
static bool sizes_missing(const uint32_t *size, size_t n) { for (size_t i = 0; i < n; ++i) if (size[i] == 0xffffffffu || size[i] == 0) return true; return false; }

extern "C" int ugs_db_set_pair_keys(ugs_db *db, const uint32_t *label_key, const uint32_t *size)
{
  if (!db) return UGS_E_ARG;
  HIPCHK(hipSetDevice(db->device));
  const bool need_size = (db->p.pair_mask & UGS_P_MIN_SIZERATIO) || (db->p.filter_mask & UGS_F_ABSKEW);
  if (need_size && size && sizes_missing(size, db->v.nseq)) { ugs_set_error("Missing size= in a database label (-min_sizeratio / -abskew need it, label.cpp:152-161)"); return UGS_E_ARG; }
  RCCHK(upload_keys(&db->d_tkey, label_key, db->v.nseq, &db->have_tkey, db->stream));
  RCCHK(upload_keys(&db->d_tsize, size, db->v.nseq, &db->have_tsize, db->stream));
  db->v.t_key = db->d_tkey; db->v.t_size = db->d_tsize;
  return UGS_OK;
}

extern "C" int ugs_batch_set_pair_keys(ugs_batch *b, const uint32_t *label_key, const uint32_t *size)
{
  if (!b) return UGS_E_ARG;
  ugs_db *db = b->db;
  HIPCHK(hipSetDevice(db->device));
  const bool need_size = (db->p.pair_mask & UGS_P_MIN_SIZERATIO) || (db->p.filter_mask & UGS_F_ABSKEW);
  if (need_size && size && sizes_missing(size, b->nq)) { ugs_set_error("Missing size= in a query label (-min_sizeratio / -abskew need it, label.cpp:152-161)"); return UGS_E_ARG; }
  if (b->d_qkey && label_key) { HIPCHK(ugs_free(b->d_qkey)); b->d_qkey = nullptr; }     // sized for the uploaded batch
  if (b->d_qsize && size) { HIPCHK(ugs_free(b->d_qsize)); b->d_qsize = nullptr; }
  RCCHK(upload_keys(&b->d_qkey, label_key, b->nq, &b->have_qkey, db->stream));
  RCCHK(upload_keys(&b->d_qsize, size, b->nq, &b->have_qsize, db->stream));
  b->v.q_key = b->d_qkey; b->v.q_size = b->d_qsize;
  return UGS_OK;
}

extern "C" int ugs_db_stats(const ugs_db *db, uint64_t *n_postings, uint64_t *n_slots, uint64_t *hbm_bytes)
{
  if (!db) return UGS_E_ARG;
  if (n_postings) *n_postings = db->n_postings;
  if (n_slots) *n_slots = db->v.slots;
  if (hbm_bytes) *hbm_bytes = db->hbm_bytes + (db->d_post16 ? db->post16_cap * 2 : 0);
  return UGS_OK;
}

// test/debug introspection (not in ugs.h): copy masked letters and the CSR index back to the host
extern "C" int ugs_db_debug_fetch(const ugs_db *db, char *masked, uint64_t *row_off, uint32_t *postings)
{
  if (!db) return UGS_E_ARG;
  HIPCHK(hipSetDevice(db->device));
  uint64_t nl = 0;
  HIPCHK(hipMemcpy(&nl, db->d_offs + db->v.nseq, 8, hipMemcpyDeviceToHost));
  if (masked && nl) HIPCHK(hipMemcpy(masked, db->d_seqs, nl, hipMemcpyDeviceToHost));
  if (row_off) HIPCHK(hipMemcpy(row_off, db->d_row_off, ((size_t)db->v.slots + 1) * 8, hipMemcpyDeviceToHost));
  if (postings && db->n_postings) HIPCHK(hipMemcpy(postings, db->d_postings, db->n_postings * 4, hipMemcpyDeviceToHost));
  return UGS_OK;
}

extern "C" int ugs_db_masked_letters(const ugs_db *db, char *out)
{
  if (!db || !out) { ugs_set_error("null argument"); return UGS_E_ARG; }
  return ugs_db_debug_fetch(db, out, nullptr, nullptr);
}

// ---------------------------------------------------------------- batch
extern "C" void ugs_batch_destroy(ugs_batch *b)
{
  if (!b) return;
  (void)hipSetDevice(b->db->device);
  // (as ugs_db_destroy: the handle's own streams and events only)
  if (b->db->stream) (void)hipStreamSynchronize(b->db->stream);
  if (b->copy_stream) (void)hipStreamSynchronize(b->copy_stream);
  if (b->ev_done) (void)hipEventSynchronize(b->ev_done);
  if (b->ev_up) (void)hipEventSynchronize(b->ev_up);
  (void)hipDeviceSynchronize();        // (device-wide, as ugs_db_destroy says)
  (void)ugs_free(b->d_qseqs); (void)ugs_free(b->d_qoffs); (void)ugs_free(b->d_cand); (void)ugs_free(b->d_cand_cnt); (void)ugs_free(b->d_cand_n);
  (void)ugs_free(b->d_hit_n); (void)ugs_free(b->d_cigar); (void)ugs_free(b->d_runs); (void)ugs_free(b->d_hits); (void)ugs_free(b->d_emit); (void)ugs_free(b->d_tb);
  (void)ugs_free(b->d_unit_ns); (void)ugs_free(b->d_unit_slots); (void)ugs_free(b->d_defer); (void)ugs_free(b->d_defer2); (void)ugs_free(b->d_qpk);
  (void)ugs_free(b->d_qkey); (void)ugs_free(b->d_qsize);
  (void)ugs_free(b->d_qthr); (void)ugs_free(b->d_ltb); (void)ugs_free(b->d_lrow); (void)ugs_free(b->d_lruns);
  (void)ugs_free(b->d_cigar_used); (void)ugs_free(b->d_ctr);
  (void)ugs_free(b->d_walk_state); (void)ugs_free(b->d_open_list); (void)ugs_free(b->d_deepU); (void)ugs_free(b->d_deepR); (void)ugs_free(b->d_keyn); (void)ugs_free(b->d_koff);
  (void)ugs_free(b->d_keys); (void)ugs_free(b->d_keys_sorted); (void)ugs_free(b->d_sort_tmp); (void)ugs_free(b->d_xpool); (void)ugs_free(b->d_xnext); (void)ugs_free(b->d_xblocks_used);
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev0m) (void)hipEventDestroy(b->ev0m);
  if (b->ev0s) (void)hipEventDestroy(b->ev0s);
  if (b->ev0r) (void)hipEventDestroy(b->ev0r);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  if (b->ev2) (void)hipEventDestroy(b->ev2);
  if (b->ev_up) (void)hipEventDestroy(b->ev_up);
  if (b->ev_done) (void)hipEventDestroy(b->ev_done);
  if (b->copy_stream) (void)hipStreamDestroy(b->copy_stream);
  if (b->h_rel) (void)hipHostFree(b->h_rel);
  delete b;
}

extern "C" int ugs_batch_create(ugs_db *db, uint32_t max_queries, uint64_t max_letters, ugs_batch **out)
{
  if (!db || !out) return UGS_E_ARG;
  HIPCHK(hipSetDevice(db->device));
  ugs_batch *b = new ugs_batch();
  memset(b, 0, sizeof(*b));
  b->db = db; b->max_queries = max_queries; b->max_letters = max_letters;
  b->nstrand = db->p.strand_both ? 2 : 1;
  b->K = (db->v.align_flags & UGS_A_DEEP) ? (uint32_t)UGS_KMAX : (uint32_t)(db->p.max_accepts + db->p.max_rejects - 1);
  // small path + pair filters: passed-over pairs are not counted (searcher.cpp:63-67), the walk can go deeper
  // (only -selfid is left to the aligner on that path: the other pair filters are applied where candidates are chosen)
  if ((db->p.pair_mask & UGS_P_SELFID) && !db->v.big) b->K = std::min<uint32_t>(UGS_KMAX, b->K + 32);
  const uint64_t units = (uint64_t)max_queries * b->nstrand;
  b->hit_slots = (uint32_t)db->v.max_accepts * (db->p.local ? db->p.max_hsps : 1u);
  int rc = UGS_OK;
#define BCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ugs_set_error("%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); ugs_batch_destroy(b); return UGS_E_HIP; } } while (0)
  BCHK(ugs_malloc(&b->d_qseqs, max_letters ? max_letters : 16));
  BCHK(ugs_malloc(&b->d_qoffs, ((size_t)max_queries + 1) * 8));
  BCHK(ugs_malloc(&b->d_cand, std::max<uint64_t>(units * b->K, 1) * 4));
  BCHK(ugs_malloc(&b->d_cand_cnt, std::max<uint64_t>(units * b->K, 1) * 4));
  BCHK(ugs_malloc(&b->d_cand_n, std::max<uint64_t>(units, 1) * 4));
  BCHK(ugs_malloc(&b->d_hit_n, std::max<uint64_t>(units, 1) * 4));
  BCHK(ugs_malloc(&b->d_defer, std::max<uint64_t>(units, 1) * 4));
  BCHK(ugs_malloc(&b->d_hits, std::max<uint64_t>(units * b->hit_slots, 1) * sizeof(ugs_hit)));
  b->cigar_cap = units * std::max(db->p.max_accepts, 1) * 12 + 4096;
  if (db->p.local) BCHK(ugs_malloc(&b->d_qthr, std::max<uint64_t>(max_queries, 1) * sizeof(int2)));
  BCHK(ugs_malloc(&b->d_cigar, b->cigar_cap * 4));
  BCHK(ugs_malloc(&b->d_qn, std::max<uint64_t>(max_queries, 1) * 4));
  BCHK(ugs_malloc(&b->d_qoff, ((uint64_t)max_queries + 1) * 4));
  b->compact_alloc = std::max<uint64_t>(units * b->hit_slots, 1);
  BCHK(ugs_malloc(&b->d_compact, b->compact_alloc * sizeof(ugs_hit)));
  if (db->v.align_flags & UGS_A_DEEP) {
    BCHK(ugs_malloc(&b->d_walk_state, std::max<uint64_t>(units, 1) * sizeof(UgsWalkState)));
    BCHK(ugs_malloc(&b->d_open_list, std::max<uint64_t>(units, 1) * 4));
    BCHK(ugs_malloc(&b->d_xblocks_used, 8));
    BCHK(hipMemset(b->d_xblocks_used, 0, 8));
  }
  b->scan_tmp_bytes = ugs_compact_tmp_bytes(max_queries);
  BCHK(ugs_malloc(&b->d_scan_tmp, b->scan_tmp_bytes));
  BCHK(ugs_malloc(&b->d_cigar_used, 8));
  BCHK(hipMemset(b->d_cigar_used, 0, 8));
  BCHK(ugs_malloc(&b->d_ctr, UGS_CTR_N * 8));
  BCHK(hipEventCreate(&b->ev0m));
  BCHK(hipEventCreate(&b->ev0)); BCHK(hipEventCreate(&b->ev0s)); BCHK(hipEventCreate(&b->ev0r)); BCHK(hipEventCreate(&b->ev1)); BCHK(hipEventCreate(&b->ev2));
  BCHK(hipEventCreateWithFlags(&b->ev_up, hipEventDisableTiming));
  BCHK(hipEventCreateWithFlags(&b->ev_done, hipEventDisableTiming));
  BCHK(hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking));
  BCHK(hipHostMalloc((void **)&b->h_rel, ((size_t)max_queries + 1) * 8, hipHostMallocDefault));
#undef BCHK
  (void)rc;
  *out = b;
  return UGS_OK;
}

// launch geometry and scratch of k_local (ugs_local.hip)
static int plan_local(ugs_batch *b)
{
  ugs_db *db = b->db;
  UgsLocalView &lv = b->lv;
  lv = db->lv;
  const uint32_t sideq = std::min<uint32_t>(b->max_qlen, 4096), sidet = std::min<uint32_t>(db->max_tlen, 4096);
  lv.seed_cap = std::max<uint32_t>(1024, ((b->max_qlen + 15u) & ~15u) + 64);
  lv.hit_slots = b->hit_slots;
  lv.tb_cap = ((unsigned long long)(sideq + 2) * (sidet + 4) + 256 + 63) & ~63ull;
  lv.rows_cap = sideq + 4;
  lv.runbuf_cap = b->max_qlen + db->max_tlen + 8;
  const size_t wave_lds = ugs_local_wave_lds(lv.W, b->max_qlen, db->max_tlen, lv.seed_cap);
  int wpb = 4;
  while (wpb > 1 && 2560 + wpb * wave_lds > LDS_MAX) wpb >>= 1;
  if (2560 + wpb * wave_lds > LDS_MAX) { ugs_set_error("usearch_local LDS footprint %zu exceeds 160 KiB (sequences too long)", 2560 + wave_lds); return UGS_E_ENVELOPE; }
  const size_t lds = 2560 + wpb * wave_lds;
  int per_cu = std::max(1, std::min(ugs_local_blocks_per_cu(64 * wpb, lds), 8));
  const uint64_t units = (uint64_t)b->nq * b->nstrand;
  uint64_t waves = std::max<uint64_t>(1, std::min<uint64_t>(units, (uint64_t)db->num_cu * per_cu * wpb));
  const uint64_t tb_budget = 24ull << 30;                       // HBM the traceback scratch may take
  waves = std::max<uint64_t>(1, std::min<uint64_t>(waves, tb_budget / lv.tb_cap));
  b->lgrid = (int)((waves + wpb - 1) / wpb); b->lwpb = wpb; b->llds = lds;
  const uint64_t nw = (uint64_t)b->lgrid * wpb;
  if (!b->d_ltb || nw * lv.tb_cap > b->ltb_alloc) {
    if (b->d_ltb) HIPCHK(ugs_free(b->d_ltb));
    HIPCHK(ugs_malloc(&b->d_ltb, nw * lv.tb_cap)); b->ltb_alloc = nw * lv.tb_cap;
  }
  if (!b->d_lrow || nw * lv.rows_cap > b->lrow_alloc) {
    if (b->d_lrow) HIPCHK(ugs_free(b->d_lrow));
    HIPCHK(ugs_malloc(&b->d_lrow, nw * lv.rows_cap * sizeof(uint2))); b->lrow_alloc = nw * lv.rows_cap;
  }
  if (!b->d_lruns || nw * 3 * lv.runbuf_cap > b->lruns_alloc) {
    if (b->d_lruns) HIPCHK(ugs_free(b->d_lruns));
    HIPCHK(ugs_malloc(&b->d_lruns, nw * 3 * lv.runbuf_cap * 4)); b->lruns_alloc = nw * 3 * lv.runbuf_cap;
  }
  lv.tb = b->d_ltb; lv.rowinfo = b->d_lrow; lv.runbuf = b->d_lruns; lv.qthr = b->d_qthr;
  return UGS_OK;
}

static std::atomic<uint64_t> g_emit_regrows{0};
extern "C" uint64_t ugs_debug_emit_regrows(void) { return g_emit_regrows.load(); }    // diagnostic: searches re-run with a larger candidate buffer

// keys per workgroup of k_rank's candidate buffer: what earlier searches of this batch object demanded (emit_limit), never more than the
// worst case.  The buffer is cut into wpb equal per-wave shares and partitions go to waves round-robin, so ONE wave may emit every
// posting of a unit's longest rows (np < wpb, or a family of near-identical targets inside one partition): the bound is per WAVE, i.e.
// wpb times that per workgroup.  One rule for the first sizing and for the regrow in ugs_batch_sync.
static uint64_t emit_cap_for(const ugs_batch *b, uint32_t ns_max)
{
  const ugs_db *db = b->db;
  const uint64_t worst = ((uint64_t)ns_max * db->max_row + 1) * (uint64_t)std::max(1, b->rl.wpb);
  return std::min<uint64_t>(worst, b->emit_limit) + (db->tune.emit_limit ? 0 : (uint64_t)db->v.np * 4 * b->K) + 64;
}

static int plan_launch(ugs_batch *b)
{
  ugs_db *db = b->db;
  const ugs_params &p = db->p;
  const uint32_t maxq = (b->max_qlen + 15u) & ~15u;
  const uint32_t maxt = (db->max_tlen + 15u) & ~15u;
  // ---- ranking geometry
  const uint32_t W = (uint32_t)p.word_len;
  const uint32_t maxNu = b->max_qlen >= W ? b->max_qlen - W + 1 : 0;
  RCCHK(db_step_table(db, maxNu + 2));
  uint32_t ns_max = 1;
  if (db->v.big) { for (uint32_t nu = 1; nu <= maxNu; ++nu) { uint32_t s = db->step[nu]; ns_max = std::max(ns_max, (nu + s - 1) / s); } }
  else ns_max = std::max(1u, maxNu);
  if (ns_max > 4095) { ugs_set_error("query needs %u sampled words > 4095 (device envelope)", ns_max); return UGS_E_ENVELOPE; }
  // LDS counter width is sized for the TYPICAL query (the longest queries' sampled-row count);
  // a query needing wider counters splits each partition into sub-ranges inside the kernel
  uint32_t ns_typ = 1;
  for (uint32_t k = 0; k <= 16 && k < maxNu; ++k) {
    const uint32_t nu = maxNu - k;
    ns_typ = std::max(ns_typ, db->v.big ? (nu + db->step[nu] - 1) / db->step[nu] : nu);
  }
  int bits = ns_typ <= 15 ? 4 : (ns_typ <= 255 ? 8 : 16);
  size_t tbl_bytes = ((size_t)db->v.gsize * bits) / 8 + 256;            // + 64 dummy words per wave
  // LDS cache of the sampled rows' partition-table rows (hot configuration: <= 15 rows, 4-bit counters)
  uint32_t part_words = 0;
  // which instantiation will run (ugs_rank.hip rank_kernel): the LDS carve and the residency differ between them
  b->rl.fast8 = bits >= 8 && !db->sparse;
  // very long index rows (the longest row 16x the average or more: an abundant family of sequences shares its words):
  // sub-rows longer than a wavefront are then common and get their own path (ugs_rank.hip range_long)
  // LONG instantiations (sub-rows longer than a wavefront are handled from the batch's registers + extra chunks instead of the generic
  // row-by-row walk): whenever the longest index row averages more than 56 postings per partition, i.e. tails are common for the
  // queries that hold its word (uniform data stays below: C2 3.9 k vs 5.0 k, C4 19 k vs 25 k; cluster_fast's centroid index passes
  // it as soon as an abundant species has a few hundred centroids).  UGS_LONGROWS=0/1 overrides (tuning).
  b->rl.longrows = db->max_row > 56u * db->v.np;
  if (db->tune.longrows >= 0) b->rl.longrows = db->tune.longrows != 0;
  const int hot = ugs_rank_is_hot(db->v.big, bits, b->rl.fast8, b->rl.longrows);
  // the two Big-path 4-bit kernels address the partition table and the rows with 32-bit byte offsets (asm loads); an index that outgrows
  // them (table of 4 GiB, a row of 2^30 postings) gets the instantiations with 64-bit address arithmetic instead (UGS_WIDE_OFFSETS=1 forces them)
  b->rl.wide = (hot == 1 || hot == 2) && ((uint64_t)db->v.slots * (db->v.np + 1) * 4 >= (1ull << 32) || db->max_row >= (1u << 30) || db->tune.wide_offsets);
  const size_t fixed = ugs_rank_fixed_lds(ns_max, b->max_qlen, part_words, hot);
  int wpb = 4;
  while (wpb > 1 && fixed + wpb * tbl_bytes > LDS_MAX) wpb >>= 1;
  if (fixed + wpb * tbl_bytes > LDS_MAX && bits == 16) {
    // queries of more than 255 sampled words (the small path samples every word: a 1 500-residue protein) whose 16-bit counter table of one
    // partition does not fit beside their row lists: a table of half the size - the kernel (the same 8 / 16-bit instantiation) walks a
    // partition in sub-ranges then, as it does for any unit that needs wider counters than the batch's typical query (scan_generic)
    bits = 8;
    tbl_bytes = ((size_t)db->v.gsize * bits) / 8 + 256;
    wpb = 4;
    while (wpb > 1 && fixed + wpb * tbl_bytes > LDS_MAX) wpb >>= 1;
  }
  if (fixed + wpb * tbl_bytes > LDS_MAX) { ugs_set_error("ranking LDS footprint %zu exceeds 160 KiB", fixed + wpb * tbl_bytes); return UGS_E_ENVELOPE; }
  // (the HOT kernel keeps its selection scratch, (4 * UGS_KMAX + 8) keys, inside the counter tables at the end of the carve: with a partition
  // size forced far below the planner's the tables alone would not hold it - the allocation then covers the scratch)
  const size_t rlds = fixed + std::max<size_t>(wpb * tbl_bytes, hot == 1 ? ((size_t)4 * UGS_KMAX + 8) * 8 : 0);
  const uint64_t units = (uint64_t)b->nq * b->nstrand;
  int per_cu = ugs_rank_blocks_per_cu(64 * wpb, rlds, db->v.big, bits, b->rl.fast8, b->rl.longrows, b->rl.wide);      // real residency (VGPRs, LDS, wave slots)
  per_cu = std::max(1, std::min(per_cu, 8));
  if (db->tune.rank_wgs) per_cu = std::min(per_cu, db->tune.rank_wgs);
  b->rl.bits = bits; b->rl.wpb = wpb; b->rl.lds = rlds; b->rl.ns_max = ns_max; b->rl.part_words = part_words; b->rl.debug_sync = db->tune.debug_sync ? 1 : 0;
  b->rl.grid = (int)std::max<uint64_t>(1, std::min<uint64_t>(units, (uint64_t)db->num_cu * per_cu));
  // every target is emitted at most once per unit; bound by postings/2 as well
  // a target with count c is emitted c times (deduplicated at selection): bounded by the postings read
  // ... and by what the searches of this batch object have actually needed: the worst case (every posting of a unit's longest rows)
  // runs into tens of GB for cluster_fast's late batches, and a multi-GB hipMalloc / hipFree per growth step costs 0.1-0.9 s each
  // (2.7 s of one C3 run).  A unit that emits more sets UGS_ERR_EMIT and its demand; ugs_batch_sync grows the buffer and runs
  // the search again.
  if (!b->emit_limit) {
    b->emit_limit = 1u << 16;
    if (db->tune.emit_limit) b->emit_limit = (uint64_t)db->tune.emit_limit;      // tests: force the regrow path
  }
  const uint64_t ecap = emit_cap_for(b, ns_max);
  if (!b->d_emit || ecap * (uint64_t)b->rl.grid > b->emit_cap_alloc) {
    // (a database that grows - cluster_fast - asks for a little more with every batch: over-allocate then, a multi-GB
    // hipMalloc per batch costs more than the batch's kernels)
    const uint64_t want = ecap * (uint64_t)b->rl.grid, cap = b->d_emit ? want + want / 2 : want;
    if (b->d_emit) HIPCHK(ugs_free(b->d_emit));
    b->d_emit = nullptr;
    HIPCHK(ugs_malloc(&b->d_emit, cap * 8));
    b->emit_cap_alloc = cap;
  }
  b->v.emit_cap = ecap;
  {   // sampled rows per unit (k_rank_setup -> k_rank)
    const uint64_t need = (uint64_t)units * b->rl.ns_max;
    if (!b->d_unit_ns) HIPCHK(ugs_malloc(&b->d_unit_ns, (size_t)b->max_queries * 2 * 4));
    if (!b->d_unit_slots || need > b->unit_slots_alloc) {
      if (b->d_unit_slots) HIPCHK(ugs_free(b->d_unit_slots));
      HIPCHK(ugs_malloc(&b->d_unit_slots, (size_t)std::max<uint64_t>(need, 1) * 4));
      b->unit_slots_alloc = need;
    }
  }
  {   // packed query planes (nt searches): 8 bytes per 16 letters, written by k_rank_setup, read by k_align.  OFF unless UGS_QPK=1: measured on
      // C2 it saves k_align 0.06 ms per 1 M queries and costs k_rank_setup 0.25 ms (DESIGN section 3, K-align)
    b->qpk_stride = 0;
    const uint32_t stride = (b->max_qlen + 15u) / 16u + 4u;
    const uint64_t need = units * (uint64_t)stride * 8u;
    if (p.is_nucleo && !p.local && db->tune.qpk && need <= (1ull << 30)) {
      if (!b->d_qpk || need > b->qpk_alloc) {
        if (b->d_qpk) HIPCHK(ugs_free(b->d_qpk));
        b->d_qpk = nullptr;
        HIPCHK(ugs_malloc(&b->d_qpk, (size_t)std::max<uint64_t>(need, 8)));
        b->qpk_alloc = need;
      }
      b->qpk_stride = stride;
    }
  }
  // ---- the bitmap ranking kernel (ugs_rank2.hip) where the index and the launch allow it: dense Big-path index (part2), at most 15
  // sampled rows for the typical query (4-bit count field; a longer query is deferred per unit), uniform rows.  Units outside its
  // envelope come back through k_rank (HOT instantiation), which runs right behind it over the deferred list.
  b->r2_grid = 0;
  b->r2.gather = 0; b->r2.post16 = nullptr; b->r2.defer2 = nullptr; b->r2.hv_grid = 0; b->r2.hv_lds = 0; b->r2.force_defer = 0;
  // (k_rank2g reads a sub-row as at most 255 postings and 256 quads per partition: an index with long rows keeps k_rank there.  k_rank3g has
  // no such limit - a heavy super-partition is halved, a unit it cannot take is deferred - so a skewed protein dictionary stays with it)
  if (db->v.part2 && db->r2_gather && bits <= 8 && ns_typ <= 63 && (!b->rl.longrows || db->tune.r3 != 0) && b->K <= 64) {
    // (a sparse index gives a unit tens of count-2 targets, not hundreds: a kept-key list of 4 K, and the eleventh wave per CU that buys)
    const uint32_t kcap = db->tune.r2_kcap ? (uint32_t)db->tune.r2_kcap : std::max<uint32_t>(116u, 4u * b->K - 12u);
    b->r2.ns_max = ns_max; b->r2.G = db->v.gsize2; b->r2.np = db->v.np2; b->r2.kcap = kcap;
    b->r2.clcap = 0; b->r2.W = 0; b->r2.gather = 1;
    int wcu;
    if (db->tune.r3 != 0) {
      // k_rank3g (ugs_rank3.hip, the default): two filter passes per super-partition instead of an exact bitmap per partition
      b->r2.gather = 2;
      b->r2.W = db->tune.r3_sp ? (uint32_t)db->tune.r3_sp : 0u;           // 0: per unit, from its postings (UgsRank2Params::clcap per super-partition)
      b->r2.clcap = db->tune.r3_pps ? (uint32_t)db->tune.r3_pps : 4096u;
      b->r2.lds = (uint32_t)ugs_rank3g_lds(kcap);
      wcu = std::max(1, std::min(ugs_rank3g_blocks_per_cu(b->r2.lds), 32));
    } else {
      b->r2.lds = (uint32_t)ugs_rank2g_lds(db->v.gsize2, kcap, db->v.np2);
      wcu = std::max(1, std::min(ugs_rank2_blocks_per_cu(b->r2.lds, 1), 32));
    }
    if (db->tune.r2_waves) wcu = std::min(wcu, db->tune.r2_waves);
    b->r2_grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((units + 3) / 4, (uint64_t)db->num_cu * wcu));
  } else
  // (cluster_fast: a centroid index always has long rows - the centroids of an abundant species share their words; the bitmap kernel
  // reads a long sub-row as several chunks and the units whose window does not fit its chunk list are deferred, so it takes those too)
  if (db->v.part2 && !db->r2_gather && bits == 4 && (!b->rl.longrows || b->cl_mode) && b->K <= 64) {
    const uint32_t nsm = std::min<uint32_t>(ns_typ, 15u);
    uint32_t kcap = db->tune.r2_kcap ? (uint32_t)db->tune.r2_kcap : std::max<uint32_t>(252u, 6u * b->K);
    if (b->cl_mode) kcap = std::min<uint32_t>(std::max<uint32_t>(kcap, db->tune.r2_kcap ? 0u : b->K + 260u), 508u);   // (the CL instantiation compacts a full list - to K keys, in eight register batches - before a partition's <= 256 keys are added)
    b->r2.ns_max = ns_max; b->r2.G = db->v.gsize2; b->r2.np = db->v.np2; b->r2.kcap = kcap;
    // 16-bit postings (r6): a plain search streams the index as offsets inside the partitions - half the bytes of the dominant stream.  A
    // second copy of the postings at half their size (C2 + 0.46 GB, C4 + 2.3 GB), made on the first search plan that wants it and again
    // after the index changed; cluster_fast's index grows batch by batch and keeps the 32-bit stream.
    b->r2.post16 = nullptr;
    if (!b->cl_mode && db->tune.r2_p16 != 0 && db->v.np2 <= 512u && db->v.gsize2 <= 65536u && db->n_postings) {
      std::lock_guard<std::mutex> lk(db->post16_mu);
      if (!db->d_post16 || db->post16_gen != db->index_gen) {
        const uint64_t want = db->n_postings + 512;                 // (a chunk reads up to 256 + 3 elements past a sub-row's start)
        if (!db->d_post16 || want > db->post16_cap) {
          (void)ugs_free(db->d_post16); db->d_post16 = nullptr; db->post16_cap = 0;
          uint16_t *np16 = nullptr;
          if (ugs_malloc(&np16, want * 2) != hipSuccess) { ugs_set_error("16-bit postings: %llu bytes of device memory not available", (unsigned long long)(want * 2)); return UGS_E_NOMEM; }
          db->d_post16 = np16; db->post16_cap = want;
        }
        HIPCHK(hipMemsetAsync(db->d_post16 + db->n_postings, 0, (db->post16_cap - db->n_postings) * 2, db->stream));
        RCCHK(ugs_build_post16(db->d_postings, db->n_postings, db->v.gsize2, db->d_post16, db->stream));
        HIPCHK(hipStreamSynchronize(db->stream));
        db->post16_gen = db->index_gen;
      }
      b->r2.post16 = db->d_post16;
    }
    // the chunk list of a window: every partition takes its rows' chunks rounded up to a multiple of 4; sized for 16 partitions of
    // the typical query with room for sub-rows of two chunks (the kernel fits each unit's window to the list)
    b->r2.clcap = std::max<uint32_t>(96u, 16u * ((nsm + 4u) / 4u * 4u) + 16u);
    b->r2.W = std::max<uint32_t>(4u, std::min<uint32_t>(28u, (db->v.np2 + 3u) / 4u * 4u));
    if (b->cl_mode && b->rl.longrows) b->r2.clcap *= 2;                 // (room for the chunks of long sub-rows)
    if (db->tune.r2_clcap) b->r2.clcap = (uint32_t)db->tune.r2_clcap / 4u * 4u;
    b->r2.lds = (uint32_t)ugs_rank2_lds(db->v.gsize2, kcap, b->r2.clcap, b->cl_mode ? 1 : 0);
    if (!b->cl_mode && !db->tune.r2_kcap && !db->tune.r2_clcap) {
      // 16 waves per CU need <= 10 240 bytes of LDS per wave: a chunk list of 128 descriptors (the scan then takes a unit's partitions in
      // two or three windows) and a kept-key list of 188 (a few hundred of a million C2 units more are deferred) buy the two waves
      // that a 7 KB bitmap otherwise costs (C2: 31.5 -> 29.5 ms per 1 M queries, r5)
      const uint32_t kc = std::max<uint32_t>(188u, 4u * b->K), cc = 128u;
      const uint32_t lds_c = (uint32_t)ugs_rank2_lds(db->v.gsize2, kc, cc, 0);
      if (lds_c <= 10240u && b->r2.lds > 10240u && kc < kcap) { kcap = kc; b->r2.kcap = kc; b->r2.clcap = cc; b->r2.lds = lds_c; }
    }
    int wcu = std::max(1, std::min(ugs_rank2_blocks_per_cu(b->r2.lds, 0, b->cl_mode ? 1 : 0, b->r2.post16 ? 1 : 0), 32));
    if (db->tune.r2_waves) wcu = std::min(wcu, db->tune.r2_waves);
    // cluster_fast: the units the CL instantiation defers (a partition with more second touches than its record list: reads of abundant
    // species) go through the heavy-unit instantiation (4-bit counters over k_rank's partitions, two passes) before k_rank sees the rest
    if (b->cl_mode && db->tune.r2_hv != 0 && db->v.np <= 1024u && (db->v.gsize & 63u) == 0 && db->v.gsize <= 65536u && b->K + 64u <= b->r2.kcap) {
      const size_t hl = ugs_rank2_hv_lds(db->v.gsize, b->r2.kcap, b->r2.clcap);
      if (hl <= 64u * 1024u) {
        if (!b->d_defer2) HIPCHK(ugs_malloc(&b->d_defer2, std::max<uint64_t>((uint64_t)b->max_queries * b->nstrand, 1) * 4));
        const int hw = std::max(1, std::min(ugs_rank2_blocks_per_cu(hl, 0, 1, 0, 1), 32));
        b->r2.defer2 = b->d_defer2; b->r2.hv_lds = (uint32_t)hl; b->r2.force_defer = db->tune.r2_hv == 2 ? 1u : 0u;
        b->r2.hv_grid = (int)std::max<uint64_t>(1, std::min<uint64_t>(units, (uint64_t)db->num_cu * hw));
      }
    }
    b->r2_grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((units + 3) / 4, (uint64_t)db->num_cu * wcu));
  }
  if (p.local) return plan_local(b);
  // ---- alignment geometry
  const uint32_t hsp_cap = db->max_tlen / (uint32_t)p.hsp_word_len + 2;
  uint32_t q2 = 64; while (q2 < maxq) q2 <<= 1;
  uint32_t bsh = 0;                                          // more than 1024 HSP words (aa: 8000): a table per bucket of words (k_align)
  while (((uint32_t)db->v.hsp_words - 1u) >> bsh >= 1024u) ++bsh;
  const size_t wstart_b = b->max_qlen < 4096 ? (((size_t)(((uint32_t)db->v.hsp_words - 1u) >> bsh) * 2 + 2 + 15) & ~(size_t)15) : 0;
  const uint32_t seed_cap = 64 * UGS_MAXREPS + 128;      // one listing round always fits; rounds of 64 seeds are extended at a time
  // per-wave LDS (mirrors the carve in k_align): control block, class + score codes of both sequences, word table,
  // sorted query words, run buffers, small-hole traceback, HSPs + chain, union{seed list | DP rows + chainer scratch}
  const size_t u_region = std::max<size_t>((size_t)seed_cap * 4, std::max<size_t>(2 * ((size_t)maxt + 8) * 4, (size_t)hsp_cap * 28));
  const bool always_counting = wstart_b != 0 && bsh == 0 && (uint64_t)maxq * 3 / 2 + 16 <= u_region / 4;
  const size_t packed_b = 2 * ((((size_t)maxq / 16 + 6) * 4 + 15) & ~(size_t)15) + 2 * ((((size_t)maxt / 16 + 6) * 4 + 15) & ~(size_t)15);   // 2-bit letters
  const size_t wave_lds = (32 + ((size_t)maxq + maxt) + ((size_t)maxq + maxt + 64) + packed_b + wstart_b + (size_t)(always_counting ? maxq : q2) * 4 +
                           2 * 32 * 4 + 1024 + (size_t)hsp_cap * (16 + 4) + u_region + 15 + 16) & ~(size_t)15;
  int awpb = 4;
  while (awpb > 1 && UGS_ALIGN_HDR + awpb * wave_lds > LDS_MAX) awpb >>= 1;
  if (UGS_ALIGN_HDR + awpb * wave_lds > LDS_MAX) { ugs_set_error("alignment LDS footprint %zu exceeds 160 KiB (sequences too long)", UGS_ALIGN_HDR + wave_lds); return UGS_E_ENVELOPE; }
  const size_t alds = UGS_ALIGN_HDR + awpb * wave_lds;
  int aper_cu = ugs_align_blocks_per_cu(64 * awpb, alds, p.is_nucleo);
  aper_cu = std::max(1, std::min(aper_cu, 8));
  if (db->tune.align_wgs) aper_cu = std::min(aper_cu, db->tune.align_wgs);
  b->al.wpb = awpb; b->al.lds = alds; b->al.hsp_cap = hsp_cap; b->al.seed_cap = seed_cap;
  b->al.grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((units + awpb - 1) / awpb, (uint64_t)db->num_cu * aper_cu));
  const int waves = b->al.grid * awpb;
  const uint64_t band_eff = ((p.align_flags & UGS_A_FULLDP) || p.band == 0) ? std::max(b->max_qlen, db->max_tlen) : (uint64_t)p.band;
  const uint64_t tb_stride = ((uint64_t)(b->max_qlen + 1) * ((uint64_t)std::max(b->max_qlen, db->max_tlen) + 2 * band_eff + 4) + 63) & ~63ull;
  const uint32_t runs_stride = 2 * (b->max_qlen + db->max_tlen + 4);
  if (!b->d_tb || tb_stride * waves > b->tb_alloc) {
    if (b->d_tb) HIPCHK(ugs_free(b->d_tb));
    HIPCHK(ugs_malloc(&b->d_tb, tb_stride * waves));
    b->tb_alloc = tb_stride * waves;
  }
  if (!b->d_runs || (uint64_t)runs_stride * waves > b->runs_alloc) {
    if (b->d_runs) HIPCHK(ugs_free(b->d_runs));
    HIPCHK(ugs_malloc(&b->d_runs, (uint64_t)runs_stride * waves * 4));
    b->runs_alloc = (uint64_t)runs_stride * waves;
  }
  b->v.tb_stride = tb_stride; b->v.runs_stride = runs_stride;
  return UGS_OK;
}

extern "C" int ugs_batch_upload(ugs_batch *b, const char *qseqs, const uint64_t *qoffs, uint32_t nq)
{
  if (!b || !qoffs || (nq && !qseqs)) return UGS_E_ARG;
  ugs_db *db = b->db;
  HIPCHK(hipSetDevice(db->device));
  if (nq > b->max_queries || qoffs[nq] - qoffs[0] > b->max_letters) { ugs_set_error("batch exceeds its capacity"); return UGS_E_CAPACITY; }
  uint32_t maxl = 0;
  for (uint32_t i = 0; i < nq; ++i) {
    uint64_t L = qoffs[i + 1] - qoffs[i];
    if (L > 65535) { ugs_set_error("query %u longer than 65535 letters", i); return UGS_E_ENVELOPE; }
    if (L > maxl) maxl = (uint32_t)L;
  }
  b->nq = nq; b->max_qlen = maxl; b->q_letters = qoffs[nq] - qoffs[0];
  // The copies run on the batch's own stream and are not waited for here: ugs_batch_search orders the kernels behind
  // them with an event.  (The caller's letters must stay untouched until the next ugs_batch_sync of this batch; from
  // page-locked memory - ugs_host_register - the copy then overlaps whatever another batch is running.)
  HIPCHK(hipEventSynchronize(b->ev_up));                       // h_rel may still feed the previous upload
  // a search of this batch that was enqueued and not yet synced still reads d_qseqs / d_qoffs: the new copies queue behind it
  if (b->searched && !b->synced) HIPCHK(hipStreamWaitEvent(b->copy_stream, b->ev_done, 0));
  for (uint32_t i = 0; i <= nq; ++i) b->h_rel[i] = qoffs[i] - qoffs[0];
  if (b->q_letters) HIPCHK(hipMemcpyAsync(b->d_qseqs, qseqs + qoffs[0], b->q_letters, hipMemcpyHostToDevice, b->copy_stream));
  HIPCHK(hipMemcpyAsync(b->d_qoffs, b->h_rel, ((size_t)nq + 1) * 8, hipMemcpyHostToDevice, b->copy_stream));
  if (db->p.local && nq) {
    // the two e-value gates of LocalAligner::AlignPos as integer score thresholds (half-units), per query length:
    //   ungapped: Score < (float)GetMinUngappedRawScore(QL) rejects            (localmulti.cpp:15, localaligner.cpp:163-168)
    //   gapped:   RawScoreToEvalue(score, QL, true) > -evalue rejects; E is monotone in the score (localaligner.cpp:199-205)
    const EStats es(db->p);
    std::vector<int2> thr(nq);
    std::vector<int2> memo(maxl + 1, make_int2(0, -1));
    for (uint32_t i = 0; i < nq; ++i) {
      const uint32_t QL = (uint32_t)(qoffs[i + 1] - qoffs[i]);
      int2 &m = memo[QL];
      if (m.y < 0) {
        const double mu = 2.0 * (double)(float)es.min_ungapped(QL ? QL : 1);
        m.x = mu > 2e9 ? 2000000000 : (mu < -2e9 ? -2000000000 : (int)ceil(mu));
        int lo = 1, hi = 1 << 24;
        if (!(es.evalue(hi * 0.5, QL) <= (double)db->p.evalue)) lo = hi = 0x7fffffff;   // nothing passes
        while (lo < hi) { const int mid = lo + (hi - lo) / 2; if (es.evalue(mid * 0.5, QL) <= (double)db->p.evalue) hi = mid; else lo = mid + 1; }
        m.y = lo;
      }
      thr[i] = m;
    }
    HIPCHK(hipMemcpyAsync(b->d_qthr, thr.data(), (size_t)nq * sizeof(int2), hipMemcpyHostToDevice, b->copy_stream));
    HIPCHK(hipStreamSynchronize(b->copy_stream));              // (thr is a local)
  }
  HIPCHK(hipEventRecord(b->ev_up, b->copy_stream));
  RCCHK(plan_launch(b));
  UgsBatchView &v = b->v;
  v.qseqs = b->d_qseqs; v.qoffs = b->d_qoffs; v.nq = nq; v.nstrand = b->nstrand; v.K = b->K; v.max_qlen = maxl;
  v.cand = b->d_cand; v.cand_cnt = b->d_cand_cnt; v.cand_n = b->d_cand_n; v.emit_buf = b->d_emit;
  v.unit_ns = b->d_unit_ns; v.unit_slots = b->d_unit_slots; v.defer_list = b->d_defer; v.use_defer = 0;
  v.qpk = b->qpk_stride ? (uint2 *)b->d_qpk : nullptr; v.qpk_stride = b->qpk_stride;
  v.hits = b->d_hits; v.hit_n = b->d_hit_n; v.cigar_pool = b->d_cigar; v.cigar_cap = b->cigar_cap;
  v.cigar_used = b->d_cigar_used; v.tb = b->d_tb; v.runs = b->d_runs; v.counters = b->d_ctr;
  v.walk_state = b->d_walk_state; v.open_list = b->d_open_list; v.walk_units = nullptr; v.n_walk = 0; v.deep_keys = nullptr; v.deep_off = nullptr;
  v.xpool = b->d_xpool; v.xnext = b->d_xnext; v.xblocks_cap = b->xblocks_cap; v.xblocks_used = b->d_xblocks_used;
  b->have_qkey = b->have_qsize = false; b->v.q_key = nullptr; b->v.q_size = nullptr;     // keys belong to one uploaded batch
  b->searched = false; b->synced = false;
  return UGS_OK;
}

// blocks until the last ugs_batch_upload of this batch has arrived in HBM (the upload itself is asynchronous)
extern "C" int ugs_batch_wait_upload(ugs_batch *b)
{
  if (!b) return UGS_E_ARG;
  HIPCHK(hipSetDevice(b->db->device));
  HIPCHK(hipEventSynchronize(b->ev_up));
  return UGS_OK;
}

// every kernel, memset and event of a batch is enqueued on the HANDLE's stream (r5 had an experiment with a stream per batch whose fall-back
// branch called itself - ADVICE r05: the optimiser folded it to the batch's null work_stream field, i.e. the legacy NULL stream)
static inline hipStream_t bstream(const ugs_batch *b) { return b->db->stream; }
static int enqueue_align(ugs_batch *b)
{
  ugs_db *db = b->db;
  HIPCHK(hipMemsetAsync(b->d_cigar_used, 0, 8, bstream(b)));
  if (db->p.local) return ugs_launch_local(db->v, b->v, b->lv, b->lgrid, b->lwpb, b->llds, bstream(b));
  return ugs_launch_align(db->v, b->v, b->al, bstream(b));
}


// ---------------------------------------------------------------- deep walks (UGS_A_DEEP; kernels: ugs_deep.hip, k_align's continuation pass)
static inline bool is_deep(const ugs_batch *b) { return (b->db->v.align_flags & UGS_A_DEEP) != 0; }

// the hit table grouped by query into d_compact on stream st.  Plain searches: count, scan and copy in one go (enqueued right behind the
// alignment stage).  Deep walks: a unit may hold more hits than its slots (overflow blocks), so the total is read back first and d_compact
// grows to it - synchronous, after the continuation passes.
static int group_hits(ugs_batch *b, uint32_t query_base, hipStream_t st)
{
  if (!b->nq) return UGS_OK;
  if (!is_deep(b))
    return ugs_compact_hits(b->d_hit_n, b->d_hits, b->nq, b->nstrand, b->hit_slots, b->d_qn, b->d_qoff, b->d_compact, b->d_scan_tmp, b->scan_tmp_bytes, query_base, st);
  RCCHK(ugs_count_hits(b->d_hit_n, b->nq, b->nstrand, b->d_qn, b->d_qoff, b->d_scan_tmp, b->scan_tmp_bytes, st));
  uint32_t last_off = 0, last_n = 0;
  HIPCHK(hipMemcpyAsync(&last_off, b->d_qoff + (b->nq - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&last_n, b->d_qn + (b->nq - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const uint64_t total = (uint64_t)last_off + last_n;
  if (total > b->compact_alloc) {
    HIPCHK(ugs_free(b->d_compact)); b->d_compact = nullptr;
    b->compact_alloc = total + total / 4 + 1024;
    HIPCHK(ugs_malloc(&b->d_compact, b->compact_alloc * sizeof(ugs_hit)));
  }
  UgsXHits x; x.state = b->d_walk_state; x.pool = b->d_xpool; x.next = b->d_xnext;
  return ugs_copy_hits(b->d_hit_n, b->d_hits, b->nq, b->nstrand, b->hit_slots, b->d_qoff, b->d_compact, query_base, &x, st);
}

// The walks k_align parked (a full list of K candidates used up, no limit met): their complete sorted candidate lists, then the
// continuation pass - in chunks whose lists fit a key budget.  Synchronous; runs inside ugs_batch_sync.
static int deep_stage(ugs_batch *b)
{
  ugs_db *db = b->db;
  hipStream_t st = bstream(b);
  const uint64_t n_open = b->ctr[UGS_CTR_OPEN];
  b->deep_units = n_open; b->deep_keys_total = 0;
  if (!n_open) return UGS_OK;
  const uint64_t nseq = std::max<uint64_t>(db->v.nseq, 1);
  // ---- scratch: two words per target and workgroup
  const uint64_t scr_budget = 6ull << 30;
  const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(n_open, 256), scr_budget / (nseq * 8)));
  // (pointers and capacities are committed only when every allocation of a group succeeded - a failed multi-GB request
  //  leaves the batch without the buffers AND without a capacity that claims them, and the call returns UGS_E_NOMEM)
  if (!b->d_deepU || (uint64_t)grid * nseq > b->deep_scr_alloc) {
    const uint64_t want = (uint64_t)grid * nseq;
    (void)ugs_free(b->d_deepU); (void)ugs_free(b->d_deepR);
    b->d_deepU = b->d_deepR = nullptr; b->deep_scr_alloc = 0;
    uint32_t *u = nullptr, *r = nullptr;
    if (ugs_malloc(&u, want * 4) != hipSuccess || ugs_malloc(&r, want * 4) != hipSuccess) {
      (void)ugs_free(u); (void)ugs_free(r);
      ugs_set_error("deep walk: 2 x %llu bytes of device scratch not available", (unsigned long long)(want * 4)); return UGS_E_NOMEM;
    }
    b->d_deepU = u; b->d_deepR = r; b->deep_scr_alloc = want; b->deep_dirty = true;
  }
  if (b->deep_dirty) {           // fresh scratch, or a pass that did not reach its end: k_deep expects (and leaves) U all zero
    HIPCHK(hipMemsetAsync(b->d_deepU, 0, b->deep_scr_alloc * 4, st));
  }
  b->deep_dirty = true;          // until this stage has run to its end
  if (n_open + 1 > b->keyn_alloc) {
    const uint64_t want = n_open + n_open / 4 + 64;
    (void)ugs_free(b->d_keyn); (void)ugs_free(b->d_koff);
    b->d_keyn = nullptr; b->d_koff = nullptr; b->keyn_alloc = 0;
    uint32_t *kn = nullptr; uint64_t *ko = nullptr;
    if (ugs_malloc(&kn, want * 4) != hipSuccess || ugs_malloc(&ko, want * 8) != hipSuccess) {
      (void)ugs_free(kn); (void)ugs_free(ko);
      ugs_set_error("deep walk: %llu bytes of device memory for the list sizes not available", (unsigned long long)(want * 12)); return UGS_E_NOMEM;
    }
    b->d_keyn = kn; b->d_koff = ko; b->keyn_alloc = want;
  }
  UgsDeepArgs a;
  a.units = b->d_open_list; a.n_units = (uint32_t)n_open; a.U = b->d_deepU; a.R = b->d_deepR; a.stride = nseq;
  a.key_n = b->d_keyn; a.key_off = nullptr; a.keys = nullptr; a.ns_max = b->rl.ns_max; a.mode = 0;
  RCCHK(ugs_launch_deep(db->v, b->v, a, grid, st));
  std::vector<uint32_t> keyn(n_open);
  HIPCHK(hipMemcpyAsync(keyn.data(), b->d_keyn, n_open * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const uint64_t key_budget = 1ull << 28;                             // keys per chunk (2 GiB + as much for the sorted copy)
  std::vector<uint64_t> koff;
  for (uint64_t lo = 0; lo < n_open; ) {
    uint64_t hi = lo, total = 0;
    koff.assign(1, 0);
    while (hi < n_open && (hi == lo || total + keyn[hi] <= key_budget)) { total += keyn[hi]; koff.push_back(total); ++hi; }
    b->deep_keys_total += total;
    if (total > b->keys_alloc) {
      const uint64_t want = total + total / 8 + 1024;
      (void)ugs_free(b->d_keys); (void)ugs_free(b->d_keys_sorted);
      b->d_keys = b->d_keys_sorted = nullptr; b->keys_alloc = 0;
      uint64_t *k1 = nullptr, *k2 = nullptr;
      if (ugs_malloc(&k1, want * 8) != hipSuccess || ugs_malloc(&k2, want * 8) != hipSuccess) {
        (void)ugs_free(k1); (void)ugs_free(k2);
        ugs_set_error("deep walk: 2 x %llu bytes of device memory for the candidate lists not available", (unsigned long long)(want * 8)); return UGS_E_NOMEM;
      }
      b->d_keys = k1; b->d_keys_sorted = k2; b->keys_alloc = want;
    }
    HIPCHK(hipMemcpyAsync(b->d_koff, koff.data(), koff.size() * 8, hipMemcpyHostToDevice, st));
    a.units = b->d_open_list + lo; a.n_units = (uint32_t)(hi - lo); a.key_n = b->d_keyn + lo; a.key_off = b->d_koff; a.keys = b->d_keys; a.mode = 1;
    RCCHK(ugs_launch_deep(db->v, b->v, a, grid, st));
    RCCHK(ugs_deep_sort(b->d_keys, b->d_keys_sorted, total, (uint32_t)(hi - lo), b->d_koff, &b->d_sort_tmp, &b->sort_tmp_bytes, st));
    // ---- the continuation pass over this chunk; a path pool or an overflow hit pool that runs out is grown and the pass repeated
    unsigned long long cig0 = 0, xb0 = 0, ctr0[UGS_CTR_N];   // (ctr0: a repeated pass must not count its pairs and hits twice)
    HIPCHK(hipMemcpyAsync(ctr0, b->d_ctr, UGS_CTR_N * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&cig0, b->d_cigar_used, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&xb0, b->d_xblocks_used, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int attempt = 0; ; ++attempt) {
      UgsBatchView v = b->v;
      v.walk_units = b->d_open_list + lo; v.n_walk = (uint32_t)(hi - lo); v.deep_keys = b->d_keys_sorted; v.deep_off = b->d_koff;
      v.xpool = b->d_xpool; v.xnext = b->d_xnext; v.xblocks_cap = b->xblocks_cap;
      const unsigned long long zero = 0;
      HIPCHK(hipMemcpyAsync(b->d_ctr + UGS_CTR_NEXT_UNIT, &zero, 8, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(b->d_ctr + UGS_CTR_ERR, &zero, 8, hipMemcpyHostToDevice, st));
      if (db->p.local) RCCHK(ugs_launch_local(db->v, v, b->lv, b->lgrid, b->lwpb, b->llds, st));
      else RCCHK(ugs_launch_align(db->v, v, b->al, st));
      unsigned long long cig1 = 0, xb1 = 0, err = 0;
      HIPCHK(hipMemcpyAsync(&cig1, b->d_cigar_used, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(&xb1, b->d_xblocks_used, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(&err, b->d_ctr + UGS_CTR_ERR, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      const bool cig_over = cig1 > b->cigar_cap, x_over = (err & UGS_ERR_XHITS) != 0;
      if (err & ~(unsigned long long)UGS_ERR_XHITS) {
        ugs_set_error("device envelope exceeded in the continuation of a deep walk (flags 0x%llx)", err);
        return UGS_E_ENVELOPE;
      }
      if (!cig_over && !x_over) break;
      if (attempt >= 4) { ugs_set_error("deep walk: pools still too small after %d passes", attempt + 1); return UGS_E_CAPACITY; }
      if (cig_over) {            // the paths written before this chunk stay: copy them into the larger pool
        const uint64_t cap = cig1 + cig1 / 4 + 4096;
        uint32_t *np = nullptr;
        HIPCHK(ugs_malloc(&np, cap * 4));
        if (cig0) HIPCHK(hipMemcpyAsync(np, b->d_cigar, cig0 * 4, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(ugs_free(b->d_cigar));
        b->d_cigar = np; b->cigar_cap = cap; b->v.cigar_pool = np; b->v.cigar_cap = cap;
      }
      if (x_over) {
        const uint32_t cap = (uint32_t)std::min<unsigned long long>(xb1 + xb1 / 4 + 64, 0xfffffff0ull);
        ugs_hit *np = nullptr; uint32_t *nn = nullptr;
        HIPCHK(ugs_malloc(&np, (size_t)cap * UGS_XBLOCK * sizeof(ugs_hit)));
        HIPCHK(ugs_malloc(&nn, (size_t)cap * 4));
        if (xb0) { HIPCHK(hipMemcpyAsync(np, b->d_xpool, (size_t)xb0 * UGS_XBLOCK * sizeof(ugs_hit), hipMemcpyDeviceToDevice, st)); HIPCHK(hipMemcpyAsync(nn, b->d_xnext, (size_t)xb0 * 4, hipMemcpyDeviceToDevice, st)); }
        HIPCHK(hipStreamSynchronize(st));
        if (b->d_xpool) HIPCHK(ugs_free(b->d_xpool));
        if (b->d_xnext) HIPCHK(ugs_free(b->d_xnext));
        b->d_xpool = np; b->d_xnext = nn; b->xblocks_cap = cap; b->v.xpool = np; b->v.xnext = nn; b->v.xblocks_cap = cap;
      }
      // (the pass is repeatable: it reads a unit's parked counters and writes only the head of its overflow chain and its hit count)
      HIPCHK(hipMemcpyAsync(b->d_cigar_used, &cig0, 8, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(b->d_xblocks_used, &xb0, 8, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(b->d_ctr, ctr0, UGS_CTR_N * 8, hipMemcpyHostToDevice, st));
      HIPCHK(hipStreamSynchronize(st));
    }
    lo = hi;
  }
  b->deep_dirty = false;
  HIPCHK(hipMemcpy(b->ctr, b->d_ctr, UGS_CTR_N * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&b->cigar_used_host, b->d_cigar_used, 8, hipMemcpyDeviceToHost));
  b->ctr[UGS_CTR_OPEN] = n_open;
  return UGS_OK;
}

extern "C" int ugs_batch_search(ugs_batch *b)
{
  if (!b) return UGS_E_ARG;
  ugs_db *db = b->db;
  HIPCHK(hipSetDevice(db->device));
  if (b->nq) {
    const bool need_key = (db->p.pair_mask & (UGS_P_SELF | UGS_P_NOTSELF)) != 0;
    const bool need_size = (db->p.pair_mask & UGS_P_MIN_SIZERATIO) || (db->p.filter_mask & UGS_F_ABSKEW);
    if ((need_key && !(db->have_tkey && b->have_qkey)) || (need_size && !(db->have_tsize && b->have_qsize))) {
      ugs_set_error("the active pair filters need ugs_db_set_pair_keys and ugs_batch_set_pair_keys (labels%s)", need_size ? " and sizes" : "");
      return UGS_E_ARG;
    }
  }
  b->v.K = b->K;
  // UGS_SETUP_STREAM=1: the counters' reset and the unit set-up of this search run on a stream of their own - beside the kernels of the batch
  // enqueued in front (the set-up is a latency-bound kernel of dependent loads) - and the ranking kernels wait for its event
  hipStream_t s0 = (db->setup_stream && b->nq && !b->cl_mode) ? db->setup_stream : bstream(b);
  HIPCHK(hipStreamWaitEvent(s0, b->ev_up, 0));                   // the batch's letters and offsets have arrived
  HIPCHK(hipMemsetAsync(b->d_ctr, 0, UGS_CTR_N * 8, s0));
  if (is_deep(b)) HIPCHK(hipMemsetAsync(b->d_xblocks_used, 0, 8, s0));
  HIPCHK(hipEventRecord(b->ev0, s0));
  // (cluster_fast's walk records - cand_key / cl_ev - are k_rank's: the bitmap kernel is for plain searches)
  const bool r2 = b->r2_grid > 0 && (b->v.cand_key ? (b->cl_mode && !b->r2.gather && b->v.cl_ev && b->v.cl_info) : !b->v.cl_ev);
  if (b->nq) RCCHK(ugs_launch_rank(db->v, b->v, b->rl, bstream(b), b->ev0s, r2 ? &b->r2 : nullptr, b->r2_grid, b->ev0r, s0, b->ev0m));
  else { HIPCHK(hipEventRecord(b->ev0s, bstream(b))); HIPCHK(hipEventRecord(b->ev0m, bstream(b))); }
  b->r2_ran = r2 && b->nq;
  HIPCHK(hipEventRecord(b->ev1, bstream(b)));
  const bool dbg = db->tune.debug_sync != 0;             // fault isolation: finish each stage before the next
  if (dbg) { HIPCHK(hipStreamSynchronize(bstream(b))); fprintf(stderr, "[ugs] ranking stage done\n"); }
  if (b->nq) RCCHK(enqueue_align(b)); else HIPCHK(hipMemsetAsync(b->d_cigar_used, 0, 8, bstream(b)));
  HIPCHK(hipEventRecord(b->ev2, bstream(b)));
  // hits grouped by query on the device right behind the alignment stage (count, scan, gather): ugs_batch_fetch is then
  // nothing but copies, which overlap the kernels of whatever batch runs next
  if (b->nq && !is_deep(b)) RCCHK(group_hits(b, b->query_base, bstream(b)));       // (deep walks: grouped by ugs_batch_sync, behind the continuation passes)
  HIPCHK(hipEventRecord(b->ev_done, bstream(b)));
  if (dbg) { HIPCHK(hipStreamSynchronize(bstream(b))); fprintf(stderr, "[ugs] alignment stage done\n"); }
  b->searched = true; b->synced = false; b->compact_base = b->query_base;
  return UGS_OK;
}

extern "C" int ugs_batch_sync(ugs_batch *b)
{
  if (!b || !b->searched) return UGS_E_ARG;
  ugs_db *db = b->db;
  HIPCHK(hipSetDevice(db->device));
  int emit_tries = 0;
  for (int attempt = 0; attempt < 3; ++attempt) {
    // (THIS batch's search, not the handle's whole stream: another batch's search may be queued behind it - a pipeline that enqueues
    // step i + 1 before it waits for step i must not wait for both; r5)
    HIPCHK(hipEventSynchronize(b->ev_done));
    // (through the batch's own non-blocking copy stream: a copy on the null stream would wait for everything queued on the handle's stream)
    HIPCHK(hipMemcpyAsync(b->ctr, b->d_ctr, UGS_CTR_N * 8, hipMemcpyDeviceToHost, b->copy_stream));
    HIPCHK(hipMemcpyAsync(&b->cigar_used_host, b->d_cigar_used, 8, hipMemcpyDeviceToHost, b->copy_stream));
    HIPCHK(hipStreamSynchronize(b->copy_stream));
    if ((b->ctr[UGS_CTR_ERR] & UGS_ERR_EMIT) && emit_tries < 6) {     // (other flags of such a run may be consequences of the truncated lists)
      // the candidate buffer was sized by earlier demand: grow it to this search's and run the search again
      ++emit_tries; ++g_emit_regrows;
      const uint64_t wpb = (uint64_t)b->rl.wpb, demand = b->ctr[UGS_CTR_EMIT_MAX] * wpb;
      b->emit_limit = std::max<uint64_t>(b->emit_limit, demand + demand / 4 + 4096);      // (to the reported demand, not doubling blindly)
      const uint64_t ecap = emit_cap_for(b, b->rl.ns_max), want = ecap * (uint64_t)b->rl.grid;
      if (ecap <= b->v.emit_cap) {      // the buffer already holds the worst case per wave: running the search again cannot help
        ugs_set_error("candidate buffer of the ranking kernel: a wave emitted %llu keys for one unit with %llu keys per workgroup in place, which is the bound of this index (%u sampled rows x longest row %llu x %d waves)",
                      (unsigned long long)b->ctr[UGS_CTR_EMIT_MAX], (unsigned long long)b->v.emit_cap, b->rl.ns_max, (unsigned long long)db->max_row, b->rl.wpb);
        return UGS_E_ENVELOPE;
      }
      if (want > b->emit_cap_alloc) {
        HIPCHK(ugs_free(b->d_emit)); b->d_emit = nullptr;
        HIPCHK(ugs_malloc(&b->d_emit, want * 8));
        b->emit_cap_alloc = want;
      }
      b->v.emit_buf = b->d_emit; b->v.emit_cap = ecap;
      RCCHK(ugs_batch_search(b));
      --attempt;
      continue;
    }
    if (b->ctr[UGS_CTR_ERR]) {
      if (b->ctr[UGS_CTR_ERR] == UGS_ERR_LOCAL_HITS) {
        ugs_set_error("more than max_hsps = %u HSPs on one accepted target; raise ugs_params.max_hsps", db->p.max_hsps);
        return UGS_E_CAPACITY;
      }
      ugs_set_error("device envelope exceeded (flags 0x%llx: 1=sampled words 2=HSP capacity 4=path runs 8=candidate buffer 16=local scratch 32=local hit slots 64=a walk wanted more than the candidates kept per strand: -selfid on the small path)", b->ctr[UGS_CTR_ERR]);
      return UGS_E_ENVELOPE;
    }
    if (b->cigar_used_host <= b->cigar_cap) {
      if (is_deep(b)) {            // the parked walks go on; then the hit table is grouped
        RCCHK(deep_stage(b));
        RCCHK(group_hits(b, b->query_base, bstream(b)));
        HIPCHK(hipStreamSynchronize(bstream(b)));
        b->compact_base = b->query_base;
      }
      b->synced = true; return UGS_OK;
    }
    // path pool too small: grow to the demanded size and re-run the alignment stage only
    HIPCHK(ugs_free(b->d_cigar));
    b->cigar_cap = b->cigar_used_host + b->cigar_used_host / 4 + 4096;
    HIPCHK(ugs_malloc(&b->d_cigar, b->cigar_cap * 4));
    b->v.cigar_pool = b->d_cigar; b->v.cigar_cap = b->cigar_cap;
    unsigned long long keep = b->ctr[UGS_CTR_POSTINGS];
    HIPCHK(hipMemsetAsync(b->d_ctr, 0, UGS_CTR_N * 8, bstream(b)));
    HIPCHK(hipMemcpyAsync(b->d_ctr, &keep, 8, hipMemcpyHostToDevice, bstream(b)));
    if (is_deep(b)) HIPCHK(hipMemsetAsync(b->d_xblocks_used, 0, 8, bstream(b)));
    RCCHK(enqueue_align(b));
    HIPCHK(hipEventRecord(b->ev2, bstream(b)));
    if (!is_deep(b)) RCCHK(group_hits(b, b->query_base, bstream(b)));
    HIPCHK(hipEventRecord(b->ev_done, bstream(b)));
  }
  ugs_set_error("path pool overflow persisted");
  return UGS_E_CAPACITY;
}

// sort.h:85-117 QuickSortOrderRecurse<float, Desc=true> (HitMgr::Sort, hitmgr.cpp:477-483)
void ugs_qs_order_desc(const float *V, int left, int right, unsigned *Order)
{
  int i = left, j = right;
  float pivot = V[Order[(left + right) / 2]];
  while (i <= j) {
    while (V[Order[i]] > pivot) i++;
    while (V[Order[j]] < pivot) j--;
    if (i <= j) { std::swap(Order[i], Order[j]); i++; j--; }
  }
  if (left < j) ugs_qs_order_desc(V, left, j, Order);
  if (i < right) ugs_qs_order_desc(V, i, right, Order);
}

// HitMgr::Sort (hitmgr.cpp:477-483) on a hit table grouped by query: each query's hits in the order of the reference's
// QuickSortOrderDesc over AlignResult::GetScore (arscorer.cpp:818-824: float fractional identity; local: float raw score)
extern "C" int ugs_hits_sort(ugs_hit *hits, const uint32_t *nhits_per_query, uint32_t nq, int local)
{
  if ((!hits && nq) || !nhits_per_query) { ugs_set_error("null argument"); return UGS_E_ARG; }
  std::vector<ugs_hit> tmp; std::vector<float> sc; std::vector<unsigned> ord;
  uint64_t k = 0;
  for (uint32_t q = 0; q < nq; ++q) {
    const uint32_t n = nhits_per_query[q];
    if (n > 1) {
      tmp.assign(hits + k, hits + k + n); sc.resize(n); ord.resize(n);
      for (uint32_t i = 0; i < n; ++i) { sc[i] = local ? tmp[i].raw_score : (float)(tmp[i].aln_len == 0 ? 0.0 : (double)tmp[i].ids / (double)tmp[i].aln_len); ord[i] = i; }
      ugs_qs_order_desc(sc.data(), 0, (int)n - 1, ord.data());
      for (uint32_t i = 0; i < n; ++i) hits[k + i] = tmp[ord[i]];
    }
    k += n;
  }
  return UGS_OK;
}

extern "C" int ugs_batch_fetch(ugs_batch *b, ugs_hit *hits, uint64_t hits_cap, uint32_t *nhits_per_query,
                               uint32_t *cigar_pool, uint64_t cigar_cap, uint64_t *cigar_used)
{
  if (!b || !b->synced || !nhits_per_query) return UGS_E_ARG;
  ugs_db *db = b->db;
  HIPCHK(hipSetDevice(db->device));
  const uint32_t nq = b->nq, ns = b->nstrand, ma = b->hit_slots;
  // (the search has been synced: grouping and the copies run on the batch's own stream, so that they overlap whatever
  // another batch has running on the handle's stream)
  hipStream_t cs = b->copy_stream;
  if (cigar_used) *cigar_used = 0;
  if (nq == 0) return UGS_OK;
  // (grouped by query on the device by ugs_batch_search; a ugs_batch_device_results call in between regroups with its
  // own query base, so regroup then)
  if (b->compact_base != 0) {
    RCCHK(group_hits(b, 0, cs));
    b->compact_base = 0;
  }
  (void)ns; (void)ma;
  uint32_t last_off = 0, last_n = 0;
  HIPCHK(hipMemcpyAsync(&last_off, b->d_qoff + (nq - 1), 4, hipMemcpyDeviceToHost, cs));
  HIPCHK(hipMemcpyAsync(&last_n, b->d_qn + (nq - 1), 4, hipMemcpyDeviceToHost, cs));
  HIPCHK(hipMemcpyAsync(nhits_per_query, b->d_qn, (size_t)nq * 4, hipMemcpyDeviceToHost, cs));
  HIPCHK(hipStreamSynchronize(cs));
  const uint64_t total = (uint64_t)last_off + last_n;
  if (total > hits_cap || b->cigar_used_host > cigar_cap) {
    if (cigar_used) *cigar_used = b->cigar_used_host;            // the demand, so that the caller can size the run pool
    ugs_set_error("output buffers too small (%llu hits, %llu runs)", (unsigned long long)total, (unsigned long long)b->cigar_used_host);
    return UGS_E_CAPACITY;
  }
  if (total) HIPCHK(hipMemcpyAsync(hits, b->d_compact, total * sizeof(ugs_hit), hipMemcpyDeviceToHost, cs));
  if (b->cigar_used_host) HIPCHK(hipMemcpyAsync(cigar_pool, b->d_cigar, b->cigar_used_host * 4, hipMemcpyDeviceToHost, cs));
  HIPCHK(hipStreamSynchronize(cs));
  if (cigar_used) *cigar_used = b->cigar_used_host;
  // HitMgr::Sort for the (rare) queries with several hits; cigar_off keeps pointing into the pool as copied
  if (ma > 1 || ns > 1) return ugs_hits_sort(hits, nhits_per_query, nq, db->p.local);
  return UGS_OK;
}

extern "C" int ugs_batch_get_stats(ugs_batch *b, ugs_batch_stats *st)
{
  if (!b || !st || !b->synced) return UGS_E_ARG;
  HIPCHK(hipSetDevice(b->db->device));
  memset(st, 0, sizeof(*st));
  HIPCHK(hipEventElapsedTime(&st->ms_rank_setup, b->ev0, b->ev0s));
  HIPCHK(hipEventElapsedTime(&st->ms_rank, b->ev0m, b->ev1));
  HIPCHK(hipEventElapsedTime(&st->ms_align, b->ev1, b->ev2));
  HIPCHK(hipEventElapsedTime(&st->ms_total, b->ev0, b->ev2));
  st->postings = b->ctr[UGS_CTR_POSTINGS];
  st->query_letters = b->q_letters * b->nstrand;
  st->target_letters = b->ctr[UGS_CTR_TLETTERS];
  st->pairs_aligned = b->ctr[UGS_CTR_PAIRS];
  st->dp_cells = b->ctr[UGS_CTR_CELLS];
  st->hits = b->ctr[UGS_CTR_HITS];
  if (b->db->tune.phase_clocks)
    fprintf(stderr, "[ugs] rank phase clocks (sum over WGs, thread 0): setup %llu scan %llu scan-wait %llu select %llu | align: %llu %llu %llu %llu\n",
            b->ctr[UGS_CTR_T0], b->ctr[UGS_CTR_T1], b->ctr[UGS_CTR_T2], b->ctr[UGS_CTR_T3], b->ctr[UGS_CTR_T4], b->ctr[UGS_CTR_T5],
            b->ctr[UGS_CTR_T6], b->ctr[UGS_CTR_T7]),
    fprintf(stderr, "[ugs] launch: rank grid %d x %d waves, lds %zu, bits %d ns_max %u gsize %u np %u | align grid %d x %d waves, lds %zu | bitmap kernel: grid %d lds %u G %u np %u kcap %u gather %u\n", b->rl.grid, b->rl.wpb, b->rl.lds,
            b->rl.bits, b->rl.ns_max, b->db->v.gsize, b->db->v.np, b->al.grid, b->al.wpb, b->al.lds, b->r2_grid, b->r2.lds, b->r2.G, b->r2.np, b->r2.kcap, b->r2.gather);
  return UGS_OK;
}

// diagnostic: the ranking kernels this PROCESS has launched so far / the ones the library holds (bit numbers: ugs_rank.hip rank_kernel)
extern "C" int ugs_debug_rank_instances(uint64_t *seen, uint64_t *compiled)
{
  unsigned long long c = 0;
  const unsigned long long s = ugs_rank_instances_seen(&c);
  if (seen) *seen = s;
  if (compiled) *compiled = c;
  return UGS_OK;
}

extern "C" const char *ugs_debug_rank_instance_name(int bit) { return ugs_rank_instance_name(bit); }

// diagnostic: the deep-walk stage of the last synced search of this batch - walks that were parked and continued, keys of their complete lists
extern "C" int ugs_debug_deep_walks(const ugs_batch *b, uint64_t *parked_units, uint64_t *list_keys)
{
  if (!b || !b->synced) return UGS_E_ARG;
  if (parked_units) *parked_units = b->deep_units;
  if (list_keys) *list_keys = b->deep_keys_total;
  return UGS_OK;
}

// diagnostic (tests/test_gpu_paths.py): which ranking code the last synced search of this batch ran.
//   out[0] units ranked by the bitmap kernel (ugs_rank2.hip)   out[1] units it deferred to k_rank
//   out[2] k_rank instantiation: big | bits << 1 | fast8 << 8 | longrows << 9      out[3] 1 if the bitmap kernel was launched
extern "C" int ugs_debug_kernel_hits(const ugs_batch *b, uint64_t *out, int n)
{
  if (!b || !out || n < 4 || !b->synced) return UGS_E_ARG;
  out[0] = b->ctr[UGS_CTR_R2_DONE]; out[1] = b->ctr[UGS_CTR_DEFER];
  out[2] = (uint64_t)(b->db->v.big ? 1 : 0) | ((uint64_t)b->rl.bits << 1) | ((uint64_t)b->rl.fast8 << 8) | ((uint64_t)b->rl.longrows << 9) | ((uint64_t)b->rl.wide << 10);
  out[3] = b->r2_ran ? 1 : 0;
  if (n >= 6) {
    float a = 0, c = 0;
    if (b->r2_ran) { if (hipEventElapsedTime(&a, b->ev0m, b->ev0r) != hipSuccess) a = 0; if (hipEventElapsedTime(&c, b->ev0r, b->ev1) != hipSuccess) c = 0; (void)hipGetLastError(); }    // (a diagnostic: an event not yet complete reads as 0, never an error)
    out[4] = (uint64_t)(a * 1000.0f + 0.5f); out[5] = (uint64_t)(c * 1000.0f + 0.5f);
  }
  if (n >= 7) out[6] = b->ctr[UGS_CTR_GROUPED];
  if (n >= 9) out[8] = b->ctr[UGS_CTR_HV_DONE];
  if (n >= 8) out[7] = !b->r2_ran ? 0u : b->r2.gather == 2u ? 3u : b->r2.gather ? 2u : b->v.cand_key ? 4u : (b->r2.post16 ? 5u : 1u);
  return UGS_OK;
}

extern "C" int ugs_batch_candidate_k(const ugs_batch *b, uint32_t *k)
{
  if (!b || !k) return UGS_E_ARG;
  *k = b->K;
  return UGS_OK;
}

extern "C" int ugs_batch_get_candidates(ugs_batch *b, uint32_t *cand, uint32_t *cnt, uint32_t *n, uint32_t k_cap)
{
  if (!b || !b->synced || !cand || !cnt || !n) return UGS_E_ARG;
  HIPCHK(hipSetDevice(b->db->device));
  const uint64_t units = (uint64_t)b->nq * b->nstrand;
  const uint32_t K = b->K;
  if (k_cap < K) { ugs_set_error("k_cap < %u", K); return UGS_E_CAPACITY; }
  std::vector<uint32_t> c(units * K ? units * K : 1), cc(units * K ? units * K : 1);
  if (units) {
    HIPCHK(hipMemcpy(c.data(), b->d_cand, units * K * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(cc.data(), b->d_cand_cnt, units * K * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(n, b->d_cand_n, units * 4, hipMemcpyDeviceToHost));
  }
  for (uint64_t u = 0; u < units; ++u)
    for (uint32_t k = 0; k < K; ++k) { cand[u * k_cap + k] = c[u * K + k]; cnt[u * k_cap + k] = cc[u * K + k]; }
  return UGS_OK;
}

extern "C" int ugs_search_batch(ugs_db *db, const char *qseqs, const uint64_t *qoffs, uint32_t nq,
                                ugs_hit *hits, uint64_t hits_cap, uint32_t *nhits_per_query,
                                uint32_t *cigar_pool, uint64_t cigar_cap, uint64_t *cigar_used)
{
  if (!db || !qoffs) return UGS_E_ARG;
  ugs_batch *b = nullptr;
  RCCHK(ugs_batch_create(db, nq, qoffs[nq] - qoffs[0], &b));
  int rc = ugs_batch_upload(b, qseqs, qoffs, nq);
  if (rc == UGS_OK) rc = ugs_batch_search(b);
  if (rc == UGS_OK) rc = ugs_batch_sync(b);
  if (rc == UGS_OK) rc = ugs_batch_fetch(b, hits, hits_cap, nhits_per_query, cigar_pool, cigar_cap, cigar_used);
  ugs_batch_destroy(b);
  return rc;
}

extern "C" int ugs_host_register(void *ptr, uint64_t bytes)
{
  if (!ptr || !bytes) { ugs_set_error("null argument"); return UGS_E_ARG; }
  HIPCHK(hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault));
  return UGS_OK;
}

extern "C" int ugs_host_unregister(void *ptr)
{
  if (!ptr) return UGS_E_ARG;
  HIPCHK(hipHostUnregister(ptr));
  return UGS_OK;
}

// ---------------------------------------------------------------- text writers
// blast6out.cpp:27-80: for global hits qstart/qend/sstart/send are 1..QL / 1..TL (the global HSP is
// the whole of both sequences, alignresult.cpp:138-145); sstart/send swap for a rev-comp query
// (arscorer.cpp:754-757); columns 11-12 are literally "*".
extern "C" int ugs_format_blast6(const ugs_hit *h, const char *qlabel, const char *tlabel, char *buf, int cap)
{
  const double FractId = h->aln_len == 0 ? 0.0 : (double)h->ids / (double)h->aln_len;
  const double PctId = 100.0 * FractId;
  unsigned TLo = 1, THi = h->tl;
  if (h->strand) { TLo = h->tl; THi = 1; }
  return snprintf(buf, (size_t)cap, "%s\t%s\t%.1f\t%u\t%u\t%u\t%u\t%u\t%u\t%u\t*\t*\n", qlabel, tlabel, PctId,
                  h->aln_len, h->mism, h->opens, 1u, h->ql, TLo, THi);
}

// blast6out.cpp:27-80 for a usearch_local hit: 1-based HSP coordinates (query coordinates on the plus strand, target
// pair swapped for a reverse-complemented query: arscorer.cpp:688-806), e-value %.2g, bit score %.1f
extern "C" int ugs_format_blast6_local(const ugs_params *p, const ugs_hit *h, const char *qlabel, const char *tlabel, char *buf, int cap)
{
  if (!p || !h) return UGS_E_ARG;
  const double FractId = h->aln_len == 0 ? 0.0 : (double)h->ids / (double)h->aln_len;
  const double PctId = 100.0 * FractId;
  unsigned QLo = h->qlo + 1, QHi = h->qhi + 1, TLo = h->tlo + 1, THi = h->thi + 1;
  if (h->strand) { QLo = h->ql - h->qhi; QHi = h->ql - h->qlo; std::swap(TLo, THi); }
  double E = 0, Bits = 0;
  ugs_local_evalue(p, (double)h->raw_score, h->ql, &E, &Bits);
  return snprintf(buf, (size_t)cap, "%s\t%s\t%.1f\t%u\t%u\t%u\t%u\t%u\t%u\t%u\t%.2g\t%.1f\n", qlabel, tlabel, PctId,
                  h->aln_len, h->mism, h->opens, QLo, QHi, TLo, THi, E, Bits);
}

// outputuc.cpp:45-93 with CompressPath (comppath.cpp:7-48): run of n>1 prints "nC", n==1 prints "C"
extern "C" int ugs_format_uc_hit(const ugs_hit *h, const uint32_t *cigar_pool, int is_nucleo,
                                 const char *qlabel, const char *tlabel, char *buf, int cap)
{
  const double FractId = h->aln_len == 0 ? 0.0 : (double)h->ids / (double)h->aln_len;
  const double PctId = 100.0 * FractId;
  const char strand = !is_nucleo ? '.' : (h->strand ? '-' : '+');
  // columns 6-7: GetIQLo / GetITLo - 0 for a global alignment; the HSP start for a usearch_local hit (arscorer.cpp:688-716)
  const bool local = (h->flags & UGS_HIT_LOCAL) != 0;
  const unsigned iqlo = !local ? 0u : (h->strand ? h->ql - h->qhi - 1 : h->qlo), itlo = !local ? 0u : h->tlo;
  int n = snprintf(buf, (size_t)cap, "H\t%u\t%u\t%.1f\t%c\t%u\t%u\t", h->target, h->ql, PctId, strand, iqlo, itlo);
  for (uint32_t k = 0; k < h->cigar_len; ++k) {
    const uint32_t r = cigar_pool[h->cigar_off + k];
    const char op = "MDI"[r & 3];
    const uint32_t len = r >> 2;
    char *dst = n < cap ? buf + n : nullptr;
    const size_t room = n < cap ? (size_t)(cap - n) : 0;
    if (len == 1) n += snprintf(dst, room, "%c", op); else n += snprintf(dst, room, "%u%c", len, op);
  }
  n += snprintf(n < cap ? buf + n : nullptr, n < cap ? (size_t)(cap - n) : 0, "\t%s\t%s\n", qlabel, tlabel);
  return n;
}

// outputuc.cpp:10-23
extern "C" int ugs_format_uc_nohit(uint32_t ql, const char *qlabel, char *buf, int cap)
{
  return snprintf(buf, (size_t)cap, "N\t*\t%u\t*\t.\t*\t*\t*\t%s\t*\n", ql, qlabel);
}

// Device-resident, query-grouped results of the last search, for callers that move them GPU-to-GPU
// (the multi-GPU driver gathers them to rank 0 with RCCL over xGMI without touching the host):
// compact hits[n_hits] (ugs_hit, `query` offset by query_base), nhits_per_query[nq] (uint32) and the
// run pool (uint32).  Hits of a query are in discovery order (strand 0 first); ugs_batch_fetch
// additionally applies HitMgr::Sort to queries with several hits.
extern "C" int ugs_batch_set_query_base(ugs_batch *b, uint32_t query_base)
{
  if (!b) return UGS_E_ARG;
  b->query_base = query_base;
  return UGS_OK;
}

extern "C" int ugs_batch_device_results(ugs_batch *b, uint32_t query_base, void **d_hits, uint64_t *hits_bytes,
                                        void **d_nhits, uint64_t *nhits_bytes, void **d_cigar, uint64_t *cigar_bytes)
{
  if (!b || !b->synced) return UGS_E_ARG;
  ugs_db *db = b->db;
  HIPCHK(hipSetDevice(db->device));
  const uint32_t nq = b->nq, ns = b->nstrand, ma = b->hit_slots;
  uint64_t total = 0;
  if (nq) {
    if (b->compact_base != query_base) {        // (a search already grouped the table with the batch's own base: ugs_batch_set_query_base)
      RCCHK(group_hits(b, query_base, b->copy_stream));
      b->compact_base = query_base;
    }
    uint32_t last_off = 0, last_n = 0;
    HIPCHK(hipMemcpyAsync(&last_off, b->d_qoff + (nq - 1), 4, hipMemcpyDeviceToHost, b->copy_stream));
    HIPCHK(hipMemcpyAsync(&last_n, b->d_qn + (nq - 1), 4, hipMemcpyDeviceToHost, b->copy_stream));
    HIPCHK(hipStreamSynchronize(b->copy_stream));
    total = (uint64_t)last_off + last_n;
  }
  if (d_hits) *d_hits = b->d_compact;
  if (hits_bytes) *hits_bytes = total * sizeof(ugs_hit);
  if (d_nhits) *d_nhits = b->d_qn;
  if (nhits_bytes) *nhits_bytes = (uint64_t)nq * 4;
  if (d_cigar) *d_cigar = b->d_cigar;
  if (cigar_bytes) *cigar_bytes = b->cigar_used_host * 4;
  return UGS_OK;
}

// ---------------------------------------------------------------- .udb files
namespace {
#pragma pack(push, 1)
struct UdbHdr {                       // udbfile.h:18-49
  uint32_t magic1, hashed, seq_index_bits, seq_pos_bits, word_width, db_step, db_accel_pct, rfu1, rfu2, utax, end_of_row;
  uint64_t slot_count, seq_count;
  uint8_t step_prefix[8];
  char alpha[64], pattern[64];
  uint32_t magic2;
};
#pragma pack(pop)
struct SeqDbHdr { uint32_t magic1, seq_count; uint64_t seq_bytes; uint32_t label_bytes, split_count, magic2, pad; };   // seqdb.h:19-27
static_assert(sizeof(UdbHdr) == 200 && sizeof(SeqDbHdr) == 32, "file header layout");
constexpr uint32_t fourcc(char a, char b, char c, char d) { return (uint32_t)(uint8_t)a | ((uint32_t)(uint8_t)b << 8) | ((uint32_t)(uint8_t)c << 16) | ((uint32_t)(uint8_t)d << 24); }
const uint32_t UDB_MAGIC1 = fourcc('F', 'B', 'D', 'U'), UDB_MAGIC2 = fourcc('f', 'B', 'D', 'U');   // MAGIC('U','D','B','F') as stored
const uint32_t UDB_MAGIC3 = fourcc('3', 'B', 'D', 'U'), UDB_MAGIC4 = fourcc('4', 'B', 'D', 'U');
const uint32_t SEQDB_MAGIC1 = 0x5E0DB3, SEQDB_MAGIC2 = 0x5E0DB4;

struct File {
  FILE *f = nullptr;
  ~File() { if (f) fclose(f); }
};
bool rd(FILE *f, void *p, size_t n) { return n == 0 || fread(p, 1, n, f) == n; }
bool wr(FILE *f, const void *p, size_t n) { return n == 0 || fwrite(p, 1, n, f) == n; }

// parse both headers; leaves the file positioned at the label offsets
int udb_open(const char *path, File &F, UdbHdr &h, SeqDbHdr &sh, ugs_udb_info &info, std::vector<uint32_t> *sizes_out)
{
  if (!path) { ugs_set_error("null path"); return UGS_E_ARG; }
  F.f = fopen(path, "rb");
  if (!F.f) { ugs_set_error("cannot open %s", path); return UGS_E_ARG; }
  if (!rd(F.f, &h, sizeof h) || h.magic1 != UDB_MAGIC1 || h.magic2 != UDB_MAGIC2) { ugs_set_error("%s: not a .udb file", path); return UGS_E_ARG; }
  const bool nt = strcmp(h.alpha, "nt") == 0, aa = strcmp(h.alpha, "aa") == 0;
  if ((!nt && !aa) || h.hashed || h.pattern[0] || h.seq_pos_bits != 0 || h.seq_index_bits != 32 || h.db_step != 1 ||
      h.db_accel_pct != 100 || h.end_of_row || h.utax || h.word_width == 0 || h.word_width > 12) {
    ugs_set_error("%s: unsupported .udb flavour (alpha '%s', hashed %u, spaced %d, coded %u/%u, dbstep %u, dbaccel %u)", path, h.alpha,
                  h.hashed, h.pattern[0] != 0, h.seq_index_bits, h.seq_pos_bits, h.db_step, h.db_accel_pct);
    return UGS_E_ENVELOPE;
  }
  uint64_t slots = 1;
  for (uint32_t k = 0; k < h.word_width; ++k) slots *= nt ? 4 : 20;       // udbparams.cpp:235-261 unhashed dictionary
  if (slots > (1ull << 32)) { ugs_set_error("%s: word width %u too large", path, h.word_width); return UGS_E_ENVELOPE; }
  std::vector<uint32_t> sizes(slots);
  if (!rd(F.f, sizes.data(), slots * 4)) { ugs_set_error("%s: truncated (row sizes)", path); return UGS_E_ARG; }
  uint32_t m = 0;
  if (!rd(F.f, &m, 4) || m != UDB_MAGIC3) { ugs_set_error("%s: bad magic3", path); return UGS_E_ARG; }
  uint64_t np = 0;
  for (uint32_t v : sizes) np += v;
  info.is_nucleo = nt; info.word_len = h.word_width; info.slots = slots; info.n_postings = np; info.nseq = h.seq_count;
  const long rows_pos = ftell(F.f);
  if (fseek(F.f, (long)(rows_pos + np * 4), SEEK_SET) != 0 || !rd(F.f, &m, 4) || m != UDB_MAGIC4) { ugs_set_error("%s: bad magic4", path); return UGS_E_ARG; }
  if (!rd(F.f, &sh, 28) ) { ugs_set_error("%s: truncated (seqdb header)", path); return UGS_E_ARG; }
  uint32_t pad = 0;
  if (!rd(F.f, &pad, 4) || sh.magic1 != SEQDB_MAGIC1 || sh.magic2 != SEQDB_MAGIC2 || sh.seq_count != h.seq_count) { ugs_set_error("%s: bad seqdb header", path); return UGS_E_ARG; }
  info.nletters = sh.seq_bytes; info.label_bytes = sh.label_bytes;
  if (sizes_out) sizes_out->swap(sizes);
  // remember where the rows start for the caller
  info.slots = slots;
  (void)rows_pos;
  return UGS_OK;
}
}  // namespace

extern "C" int ugs_udb_stat(const char *path, ugs_udb_info *info)
{
  if (!info) { ugs_set_error("null argument"); return UGS_E_ARG; }
  memset(info, 0, sizeof *info);
  File F; UdbHdr h; SeqDbHdr sh;
  return udb_open(path, F, h, sh, *info, nullptr);
}

extern "C" int ugs_udb_read(const char *path, char *seqs, uint64_t *offs, char *labels, uint32_t *row_sizes, uint32_t *postings)
{
  File F; UdbHdr h; SeqDbHdr sh; ugs_udb_info info;
  memset(&info, 0, sizeof info);
  std::vector<uint32_t> sizes;
  int rc = udb_open(path, F, h, sh, info, &sizes);
  if (rc != UGS_OK) return rc;
  const uint64_t n = info.nseq;
  // positioned behind the seqdb header: label offsets, labels, lengths, letters
  std::vector<uint32_t> loffs(n), lens(n);
  std::vector<char> lbuf(info.label_bytes);
  if (!rd(F.f, loffs.data(), n * 4) || !rd(F.f, lbuf.data(), info.label_bytes) || !rd(F.f, lens.data(), n * 4)) { ugs_set_error("%s: truncated (labels/lengths)", path); return UGS_E_ARG; }
  if (labels) {                       // re-pack in target order (the file keeps them in order already; offsets are honoured anyway)
    uint64_t o = 0;
    for (uint64_t k = 0; k < n; ++k) {
      if (loffs[k] >= info.label_bytes) { ugs_set_error("%s: bad label offset", path); return UGS_E_ARG; }
      const size_t l = strnlen(lbuf.data() + loffs[k], info.label_bytes - loffs[k]) + 1;
      if (o + l > info.label_bytes) { ugs_set_error("%s: labels overlap", path); return UGS_E_ARG; }
      memcpy(labels + o, lbuf.data() + loffs[k], l - 1); labels[o + l - 1] = 0; o += l;
    }
  }
  uint64_t tot = 0;
  for (uint64_t k = 0; k < n; ++k) { if (offs) offs[k] = tot; tot += lens[k]; }
  if (offs) offs[n] = tot;
  if (tot != info.nletters) { ugs_set_error("%s: sequence lengths do not add up", path); return UGS_E_ARG; }
  if (seqs && !rd(F.f, seqs, tot)) { ugs_set_error("%s: truncated (letters)", path); return UGS_E_ARG; }
  if (row_sizes) memcpy(row_sizes, sizes.data(), sizes.size() * 4);
  if (postings) {
    if (fseek(F.f, (long)(sizeof(UdbHdr) + info.slots * 4 + 4), SEEK_SET) != 0 || !rd(F.f, postings, info.n_postings * 4)) { ugs_set_error("%s: truncated (rows)", path); return UGS_E_ARG; }
  }
  return UGS_OK;
}

extern "C" int ugs_udb_write(const char *path, const ugs_db *db, const char *labels, uint64_t label_bytes)
{
  if (!path || !db || (!labels && db->v.nseq)) { ugs_set_error("null argument"); return UGS_E_ARG; }
  const uint64_t n = db->v.nseq, slots = db->v.slots;
  if (label_bytes > 0xfffffbffull) { ugs_set_error("label data too big"); return UGS_E_ENVELOPE; }
  std::vector<uint32_t> loffs(n);
  {
    uint64_t o = 0;
    for (uint64_t k = 0; k < n; ++k) {
      if (o >= label_bytes) { ugs_set_error("fewer than %llu labels", (unsigned long long)n); return UGS_E_ARG; }
      loffs[k] = (uint32_t)o;
      o += strnlen(labels + o, label_bytes - o) + 1;
    }
    if (o != label_bytes) { ugs_set_error("label_bytes does not match %llu NUL-terminated labels", (unsigned long long)n); return UGS_E_ARG; }
  }
  std::vector<uint64_t> row_off(slots + 1), offs(n + 1);
  std::vector<uint32_t> postings(db->n_postings);
  HIPCHK(hipSetDevice(db->device));
  HIPCHK(hipMemcpy(offs.data(), db->d_offs, (n + 1) * 8, hipMemcpyDeviceToHost));
  std::vector<char> seqs(offs[n]);
  int rc = ugs_db_debug_fetch(db, seqs.data(), row_off.data(), postings.data());
  if (rc != UGS_OK) return rc;
  UdbHdr h; memset(&h, 0, sizeof h);                          // UDBFileHdr::FromParams udbio.cpp:13-52
  h.magic1 = UDB_MAGIC1; h.seq_index_bits = 32; h.word_width = (uint32_t)db->p.word_len; h.db_step = 1; h.db_accel_pct = 100;
  h.seq_count = n; strcpy(h.alpha, db->p.is_nucleo ? "nt" : "aa"); h.magic2 = UDB_MAGIC2;
  std::vector<uint32_t> sizes(slots), lens(n);
  for (uint64_t s = 0; s < slots; ++s) sizes[s] = (uint32_t)(row_off[s + 1] - row_off[s]);
  for (uint64_t k = 0; k < n; ++k) lens[k] = (uint32_t)(offs[k + 1] - offs[k]);
  SeqDbHdr sh; memset(&sh, 0, sizeof sh);
  sh.magic1 = SEQDB_MAGIC1; sh.seq_count = (uint32_t)n; sh.seq_bytes = offs[n]; sh.label_bytes = (uint32_t)label_bytes; sh.magic2 = SEQDB_MAGIC2;
  File F;
  F.f = fopen(path, "wb");
  if (!F.f) { ugs_set_error("cannot create %s", path); return UGS_E_ARG; }
  const uint32_t m3 = UDB_MAGIC3, m4 = UDB_MAGIC4;
  const bool ok = wr(F.f, &h, sizeof h) && wr(F.f, sizes.data(), slots * 4) && wr(F.f, &m3, 4) && wr(F.f, postings.data(), postings.size() * 4) &&
                  wr(F.f, &m4, 4) && wr(F.f, &sh, sizeof sh) && wr(F.f, loffs.data(), n * 4) && wr(F.f, labels, label_bytes) &&
                  wr(F.f, lens.data(), n * 4) && wr(F.f, seqs.data(), seqs.size());
  if (!ok) { ugs_set_error("write error on %s", path); return UGS_E_ARG; }
  return UGS_OK;
}
