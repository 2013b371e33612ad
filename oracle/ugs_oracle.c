/*
 * ugs_oracle.c - TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's usearch_global hot path.
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src).  Scores are float like the reference (all reachable values are
 * exact half-integers).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use this file; the product (usearch12_amd/) never does.
 *
 * Pinned against the compiled unmodified reference (oracle/_ref/usearch12, oracle/_ref/ref_xdrop) through the
 * golden fixtures in tests/golden/:
 *   usearch_global incl. accept filters, -fulldp, -gaforce, -hardmask   tests/test_oracle_golden.py   (manifest.json)
 *   pair filters of Accepter::RejectPair, -abskew                        tests/test_oracle_pairs.py    (pairs_manifest.json)
 *   usearch_local (AlignMulti, AlignPos, EStats, local accept rules)     tests/test_oracle_local.py    (local_manifest.json)
 *   gapped x-drop (XDropFwd/Bwd/Split/AlignMem)                          tests/test_oracle_xdrop.py    (xdrop_{nt,aa}.txt)
 *   masking + index vs the reference's .udb                              tests/test_udb.py
 */
#include "ugs_oracle.h"

#include <ctype.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned char byte;

#define INVALID_LETTER 0xffu
#define BAD_WORD 0xffffffffu
#define MINUS_INFINITY (-9e9f)          /* mx.h:12 */
#define MAXREPS 8                       /* hspfinder.h:10 */
#define TB_DM 0x01                      /* tracebit.h:4-7 */
#define TB_IM 0x02
#define TB_MD 0x04
#define TB_MI 0x08

/* ------------------------------------------------------------------ tables */

static byte g_c2l_nt[256], g_c2l_aa[256];
static byte g_match_nt[256][256], g_match_aa[256][256];
static float g_subst_nt_default[256][256];
static float g_subst_aa[256][256];
static byte g_comp[256];
static int g_tables_done = 0;
static pthread_mutex_t g_tables_lock = PTHREAD_MUTEX_INITIALIZER;

/* BLOSUM62, NCBI 1/2-bit units, standard order; values as blosum62.cpp:22-49 holds them */
static const char B62_ORDER[] = "ARNDCQEGHILKMFPSTWYVBZX*";
static const signed char B62[24][24] = {
  { 4,-1,-2,-2, 0,-1,-1, 0,-2,-1,-1,-1,-1,-2,-1, 1, 0,-3,-2, 0,-2,-1, 0,-4},
  {-1, 5, 0,-2,-3, 1, 0,-2, 0,-3,-2, 2,-1,-3,-2,-1,-1,-3,-2,-3,-1, 0,-1,-4},
  {-2, 0, 6, 1,-3, 0, 0, 0, 1,-3,-3, 0,-2,-3,-2, 1, 0,-4,-2,-3, 3, 0,-1,-4},
  {-2,-2, 1, 6,-3, 0, 2,-1,-1,-3,-4,-1,-3,-3,-1, 0,-1,-4,-3,-3, 4, 1,-1,-4},
  { 0,-3,-3,-3, 9,-3,-4,-3,-3,-1,-1,-3,-1,-2,-3,-1,-1,-2,-2,-1,-3,-3,-2,-4},
  {-1, 1, 0, 0,-3, 5, 2,-2, 0,-3,-2, 1, 0,-3,-1, 0,-1,-2,-1,-2, 0, 3,-1,-4},
  {-1, 0, 0, 2,-4, 2, 5,-2, 0,-3,-3, 1,-2,-3,-1, 0,-1,-3,-2,-2, 1, 4,-1,-4},
  { 0,-2, 0,-1,-3,-2,-2, 6,-2,-4,-4,-2,-3,-3,-2, 0,-2,-2,-3,-3,-1,-2,-1,-4},
  {-2, 0, 1,-1,-3, 0, 0,-2, 8,-3,-3,-1,-2,-1,-2,-1,-2,-2, 2,-3, 0, 0,-1,-4},
  {-1,-3,-3,-3,-1,-3,-3,-4,-3, 4, 2,-3, 1, 0,-3,-2,-1,-3,-1, 3,-3,-3,-1,-4},
  {-1,-2,-3,-4,-1,-2,-3,-4,-3, 2, 4,-2, 2, 0,-3,-2,-1,-2,-1, 1,-4,-3,-1,-4},
  {-1, 2, 0,-1,-3, 1, 1,-2,-1,-3,-2, 5,-1,-3,-1, 0,-1,-3,-2,-2, 0, 1,-1,-4},
  {-1,-1,-2,-3,-1, 0,-2,-3,-2, 1, 2,-1, 5, 0,-2,-1,-1,-1,-1, 1,-3,-1,-1,-4},
  {-2,-3,-3,-3,-2,-3,-3,-3,-1, 0, 0,-3, 0, 6,-4,-2,-2, 1, 3,-1,-3,-3,-1,-4},
  {-1,-2,-2,-1,-3,-1,-1,-2,-2,-3,-3,-1,-2,-4, 7,-1,-1,-4,-3,-2,-2,-1,-2,-4},
  { 1,-1, 1, 0,-1, 0, 0, 0,-1,-2,-2, 0,-1,-2,-1, 4, 1,-3,-2,-2, 0, 0, 0,-4},
  { 0,-1, 0,-1,-1,-1,-1,-2,-2,-1,-1,-1,-1,-2,-1, 1, 5,-2,-2, 0,-1,-1, 0,-4},
  {-3,-3,-4,-4,-2,-2,-3,-2,-2,-3,-2,-3,-1, 1,-4,-3,-2,11, 2,-3,-4,-3,-2,-4},
  {-2,-2,-2,-3,-2,-1,-2,-3, 2,-1,-1,-2,-1, 3,-3,-2,-2, 2, 7,-1,-3,-2,-1,-4},
  { 0,-3,-3,-3,-1,-2,-2,-3,-3, 3, 1,-2, 1,-1,-2,-2, 0,-3,-1, 4,-3,-2,-1,-4},
  {-2,-1, 3, 4,-3, 0, 1,-1, 0,-3,-4, 0,-3,-3,-2, 0,-1,-4,-3,-3, 4, 1,-1,-4},
  {-1, 0, 0, 1,-3, 3, 4,-2, 0,-3,-3, 1,-1,-3,-1, 0,-1,-3,-2,-2, 1, 4,-1,-4},
  { 0,-1,-1,-1,-2,-1,-1,-1,-1,-1,-1,-1,-1,-1,-2, 0, 0,-2,-1,-1,-1,-1,-1,-4},
  {-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4, 1},
};

/* setnucmx.cpp:11-99: ACGTU (either case) score match/mismatch by letter identity
 * (T==U); every other byte scores 0 (the N loop writes zeros; the '?' loop is a no-op) */
static void fill_subst_nt(float mx[256][256], float match, float mismatch)
{
  static const char alpha[] = "ACGTU";
  memset(mx, 0, sizeof(float) * 256 * 256);
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 5; ++j) {
      float v = (g_c2l_nt[(byte)alpha[i]] == g_c2l_nt[(byte)alpha[j]]) ? match : mismatch;
      byte ui = (byte)alpha[i], uj = (byte)alpha[j];
      byte li = (byte)tolower(ui), lj = (byte)tolower(uj);
      mx[ui][uj] = v; mx[ui][lj] = v; mx[li][uj] = v; mx[li][lj] = v;
    }
}

static void init_tables(void)
{
  pthread_mutex_lock(&g_tables_lock);
  if (g_tables_done) { pthread_mutex_unlock(&g_tables_lock); return; }

  /* alpha.cpp:529-790, 1309-1567 */
  memset(g_c2l_nt, 0xff, 256);
  memset(g_c2l_aa, 0xff, 256);
  g_c2l_nt['A'] = g_c2l_nt['a'] = 0; g_c2l_nt['C'] = g_c2l_nt['c'] = 1;
  g_c2l_nt['G'] = g_c2l_nt['g'] = 2; g_c2l_nt['T'] = g_c2l_nt['t'] = 3;
  g_c2l_nt['U'] = g_c2l_nt['u'] = 3;
  {
    static const char aa[] = "ACDEFGHIKLMNPQRSTVWY";
    for (int i = 0; i < 20; ++i) {
      g_c2l_aa[(byte)aa[i]] = (byte)i;
      g_c2l_aa[(byte)tolower(aa[i])] = (byte)i;
    }
  }

  /* alpha2.cpp:54-69 (IUPAC codes), :90-150 (bits) */
  byte nucbit[256], iupac[256];
  memset(nucbit, 0, 256); memset(iupac, 0, 256);
  nucbit['A'] = nucbit['a'] = 1; nucbit['C'] = nucbit['c'] = 2;
  nucbit['G'] = nucbit['g'] = 4; nucbit['T'] = nucbit['t'] = 8; nucbit['U'] = nucbit['u'] = 8;
  for (int c = 0; c < 256; ++c) iupac[c] = nucbit[c];
  {
    static const struct { char code; const char *chars; } codes[] = {
      {'M', "AC"}, {'R', "AG"}, {'W', "AT"}, {'S', "CG"}, {'Y', "CT"}, {'K', "GT"},
      {'V', "ACG"}, {'H', "ACT"}, {'D', "AGT"}, {'B', "CGT"}, {'X', "GATC"}, {'N', "GATC"}};
    for (unsigned k = 0; k < sizeof(codes) / sizeof(codes[0]); ++k) {
      byte bits = 0;
      for (const char *p = codes[k].chars; *p; ++p) bits |= nucbit[(byte)*p];
      iupac[(byte)codes[k].code] = bits;
      iupac[(byte)tolower(codes[k].code)] = bits;
    }
  }

  /* alpha2.cpp:220-300 Init_MatchMxs */
  for (int i = 0; i < 256; ++i)
    for (int j = 0; j < 256; ++j) {
      int ai = isalpha(i) != 0, aj = isalpha(j) != 0;
      if (!ai || !aj) {
        int gi = (i == '-' || i == '.'), gj = (j == '-' || j == '.');
        g_match_nt[i][j] = g_match_aa[i][j] = (byte)(gi && gj);
        continue;
      }
      if (toupper(i) == toupper(j)) { g_match_nt[i][j] = g_match_aa[i][j] = 1; continue; }
      g_match_aa[i][j] = (byte)(toupper(i) == 'X' || toupper(j) == 'X');
      int eqij = (nucbit[i] & iupac[j]) != 0;
      int eqji = (nucbit[j] & iupac[i]) != 0;
      g_match_nt[i][j] = (byte)(eqij || eqji);
    }
  g_match_aa['B']['N'] = g_match_aa['N']['B'] = 1;
  g_match_aa['B']['D'] = g_match_aa['D']['B'] = 1;
  g_match_aa['Z']['Q'] = g_match_aa['Q']['Z'] = 1;
  g_match_aa['Z']['E'] = g_match_aa['E']['Z'] = 1;

  fill_subst_nt(g_subst_nt_default, 1.0f, -2.0f);

  /* blosum62.cpp:51-84: both cases of the 24 symbols, everything else 0 */
  memset(g_subst_aa, 0, sizeof(g_subst_aa));
  for (int i = 0; i < 24; ++i)
    for (int j = 0; j < 24; ++j) {
      float v = (float)B62[i][j];
      byte ui = (byte)B62_ORDER[i], uj = (byte)B62_ORDER[j];
      byte li = (byte)tolower(ui), lj = (byte)tolower(uj);
      g_subst_aa[ui][uj] = v; g_subst_aa[ui][lj] = v; g_subst_aa[li][uj] = v; g_subst_aa[li][lj] = v;
    }

  /* alpha.cpp:3005-3265 g_CharToCompChar: '?' (= keep the char) for anything unlisted;
   * note lower-case 'u' is NOT listed */
  memset(g_comp, '?', 256);
  {
    static const char from[] = "ABCDGHKMNRSTUVWXY";
    static const char to[]   = "TVGHCDMKNYSAABWXR";
    for (int k = 0; from[k]; ++k) {
      g_comp[(byte)from[k]] = (byte)to[k];
      if (from[k] != 'U') g_comp[(byte)tolower(from[k])] = (byte)tolower(to[k]);
    }
  }
  g_tables_done = 1;
  pthread_mutex_unlock(&g_tables_lock);
}

/* ------------------------------------------------------------------ params */

void orc_params_init(ugs_params *p, int is_nucleo, double id)
{
  memset(p, 0, sizeof(*p));
  p->is_nucleo = is_nucleo;
  p->word_len = is_nucleo ? 8 : 5;           /* udbparams.cpp:235-261 */
  p->id = (float)id;
  p->id_accept = (double)(float)id;      /* options are stored as float: opts.cpp:265 */
  p->id_set = 1;
  p->strand_both = 0;
  p->max_accepts = 1;                        /* terminator.cpp:26-31 */
  p->max_rejects = 32;
  p->big = 100000;                           /* o_defaults.inc */
  p->bump_pct = 50;
  p->stepwords = 8;
  p->band = 16;
  p->minhsp = 16;
  p->xdrop_nw = 8.0f;
  p->match = 1.0f;
  p->mismatch = -2.0f;
  p->hsp_word_len = is_nucleo ? 5 : 3;       /* alnheuristics.cpp:36,44 */
  p->dbmask = 1;
  p->xdrop_u = 16.0f; p->xdrop_g = 32.0f;    /* o_defaults.inc:20,22 */
  p->local_open = -10.0f; p->local_ext = -1.0f;
  p->ka_dbsize = 1e9f;                       /* o_defaults.inc:2 */
  p->max_hsps = 8;
}

/* usearch_local: -evalue is required, -id optional (ranking falls back to 0.5, makedbsearcher.cpp:172) */
void orc_params_set_local(ugs_params *p, double evalue, int id_set)
{
  p->local = 1;
  p->evalue = (float)evalue;
  if (!id_set) { p->id = 0.5f; p->id_accept = 0.5; p->id_set = 0; }
}

/* ------------------------------------------------------------------ db */

typedef struct {
  unsigned Loi, Loj, Len;
  float Score;
} HSP;

typedef struct {
  /* gap penalties, alnparams.h */
  float OpenA, OpenB, ExtA, ExtB, LOpenA, LOpenB, LExtA, LExtB, ROpenA, ROpenB, RExtA, RExtB;
} AlnPen;

struct orc_db {
  ugs_params p;
  uint32_t nseq;
  char *seqs;           /* masked copy */
  uint64_t *offs;
  uint32_t alpha;       /* 4 / 20 */
  uint64_t slots;
  uint64_t *row_off;    /* slots+1 */
  uint32_t *postings;
  const byte *c2l;
  float (*subst)[256];
  float subst_nt_custom[256][256];
  const byte (*match)[256];
  AlnPen ap;            /* alnparams.cpp:380-384 */
  /* alnheuristics.cpp:26-62 */
  float MinGlobalHSPFractId, MinGlobalHSPScore, XDropGlobalHSP;
  unsigned MinGlobalHSPLength, BandRadius, HSPw, HSPWordCount;
  int big;              /* udbusortedsearcher.cpp:39-58 latch (DB is static here) */
  orc_stats stats;
  uint32_t maxlen;
  /* pair filters / -abskew: label keys and ;size= annotations (UINT32_MAX = none) of the DB and of the current queries */
  uint32_t *t_key, *t_size;
  const uint32_t *q_key, *q_size;
  /* cluster_fast: the index grows by one centroid at a time (UDBData::AddSIToDB_CopyData udbbuild.cpp:286-291):
   * rows are then per-slot growable arrays like the reference's m_UDBRows (udbbuild.cpp:74-128) instead of the CSR */
  uint32_t **dyn_rows; uint32_t *dyn_size, *dyn_cap;
  uint64_t seq_cap; uint32_t off_cap;
};

static inline const uint32_t *db_row(const orc_db *db, uint32_t word, uint64_t *size)
{
  if (db->dyn_rows) { *size = db->dyn_size[word]; return db->dyn_rows[word]; }
  *size = db->row_off[word + 1] - db->row_off[word];
  return db->postings + db->row_off[word];
}

/* fastmask.cpp:88-158 FastMaskSeq, soft mask (hardmask off), in place */
static void fastmask_impl(char *seq_, uint32_t L, byte Hard);
void orc_fastmask(char *seq_, uint32_t L) { fastmask_impl(seq_, L, 0); }
/* Hard != 0: -hardmask, the masked letters become Hard ('N' / 'X') instead of lower case (fastmask.cpp:98,117-122,145-150);
 * the reference masks in place (seqdb.cpp:446), so the dinucleotide pass reads the letters the first pass wrote */
static void fastmask_impl(char *seq_, uint32_t L, byte Hard)
{
  byte *Seq = (byte *)seq_;
  for (unsigned i = 0; i < L; ++i) Seq[i] = (byte)toupper(Seq[i]);
  if (L < 2) return;
  const unsigned k1 = 5, j1 = 2, k2 = 5, j2 = 1;
  byte Lastc = '?';
  unsigned Start = UINT_MAX;
  for (unsigned i = 0; i < L; ++i) {
    byte c = (byte)toupper(Seq[i]);
    if (c != Lastc || i + 1 == L) {
      unsigned n1 = i - Start;               /* unsigned wrap with Start==UINT_MAX intended */
      if (n1 >= k1)
        for (unsigned j = Start + j1; j < i; ++j) Seq[j] = Hard ? Hard : (byte)tolower(Seq[j]);
      Start = i;
    }
    Lastc = c;
  }
  for (unsigned StartPos = 0; StartPos <= 1; ++StartPos) {
    unsigned LastPair = UINT_MAX;
    unsigned Start2 = UINT_MAX;
    for (unsigned i = StartPos; i < L - 1; i += 2) {
      byte c1 = (byte)toupper(Seq[i]), c2 = (byte)toupper(Seq[i + 1]);
      unsigned Pair = ((unsigned)c1 << 8) + c2;
      if (Pair != LastPair) {
        unsigned n2 = i - Start2;
        if (n2 >= k2)
          for (unsigned j = Start2 + (Hard ? j2 : 2 * j2); j < i; ++j) Seq[j] = Hard ? Hard : (byte)tolower(Seq[j]);
        Start2 = i;
      }
      LastPair = Pair;
    }
  }
}

/* seqinfo.cpp:292-323 */
void orc_revcomp(const char *seq, uint32_t len, char *out)
{
  init_tables();
  for (uint32_t i = 0; i < len; ++i) {
    byte c = (byte)seq[i];
    byte rc = g_comp[c];
    if (rc == '?') rc = c;
    out[len - i - 1] = (char)rc;
  }
}

/* udbparams.cpp:540-555 SeqToWordNoPattern */
static inline uint32_t seq_to_word(const orc_db *db, const byte *s)
{
  uint32_t w = 0;
  for (int i = 0; i < db->p.word_len; ++i) {
    byte c = s[i];
    if (c >= 'a' && c <= 'z') return BAD_WORD;   /* islower, C locale */
    unsigned l = db->c2l[c];
    if (l == INVALID_LETTER) return BAD_WORD;
    w = w * db->alpha + l;
  }
  return w;
}

/* udbbuild.cpp:303-398 FromSeqDB + :256-284 AddSeqNoncoded: each target once per
 * distinct valid word, rows ascending by target index */
static int build_index(orc_db *db)
{
  uint64_t slots = 1;
  for (int i = 0; i < db->p.word_len; ++i) slots *= db->alpha;
  db->slots = slots;
  db->row_off = (uint64_t *)calloc(slots + 1, sizeof(uint64_t));
  uint32_t *stamp = (uint32_t *)calloc(slots, sizeof(uint32_t));
  if (!db->row_off || !stamp) return UGS_E_NOMEM;
  const int W = db->p.word_len;
  for (int pass = 0; pass < 2; ++pass) {
    uint64_t *fill = NULL;
    if (pass == 1) {
      uint64_t tot = 0;
      for (uint64_t s = 0; s < slots; ++s) { uint64_t c = db->row_off[s + 1]; db->row_off[s + 1] = tot; tot += c; }
      /* now row_off[s+1] = start of row s; shift */
      for (uint64_t s = 0; s < slots; ++s) db->row_off[s] = db->row_off[s + 1];
      db->row_off[slots] = tot;
      db->postings = (uint32_t *)malloc((tot ? tot : 1) * sizeof(uint32_t));
      fill = (uint64_t *)malloc(slots * sizeof(uint64_t));
      if (!db->postings || !fill) return UGS_E_NOMEM;
      memcpy(fill, db->row_off, slots * sizeof(uint64_t));
      memset(stamp, 0, slots * sizeof(uint32_t));
    }
    for (uint32_t t = 0; t < db->nseq; ++t) {
      const byte *s = (const byte *)db->seqs + db->offs[t];
      uint64_t L = db->offs[t + 1] - db->offs[t];
      if (L < (uint64_t)W) continue;
      for (uint64_t pos = 0; pos + W <= L; ++pos) {
        uint32_t w = seq_to_word(db, s + pos);
        if (w == BAD_WORD) continue;
        if (stamp[w] == t + 1) continue;
        stamp[w] = t + 1;
        if (pass == 0) db->row_off[w + 1]++;
        else db->postings[fill[w]++] = t;
      }
    }
    if (pass == 1) free(fill);
    else {
      /* after pass 0 row_off[s+1] holds the size of row s */
    }
  }
  free(stamp);
  return UGS_OK;
}

int orc_db_create(const ugs_params *p, const char *seqs, const uint64_t *offs, uint32_t nseq,
                  orc_db **out)
{
  init_tables();
  orc_db *db = (orc_db *)calloc(1, sizeof(orc_db));
  if (!db) return UGS_E_NOMEM;
  db->p = *p;
  db->nseq = nseq;
  uint64_t tot = offs[nseq];
  db->seqs = (char *)malloc(tot ? tot : 1);
  db->offs = (uint64_t *)malloc((nseq + 1) * sizeof(uint64_t));
  memcpy(db->seqs, seqs, tot);
  memcpy(db->offs, offs, (nseq + 1) * sizeof(uint64_t));
  db->alpha = p->is_nucleo ? 4 : 20;
  db->c2l = p->is_nucleo ? g_c2l_nt : g_c2l_aa;
  db->match = p->is_nucleo ? (const byte(*)[256])g_match_nt : (const byte(*)[256])g_match_aa;
  if (p->is_nucleo) {
    fill_subst_nt(db->subst_nt_custom, p->match, p->mismatch);
    db->subst = db->subst_nt_custom;
  } else
    db->subst = g_subst_aa;
  /* makeudb.cpp:11-25 MaskDB -> seqdb.cpp:415 Mask (in place) */
  uint32_t maxlen = 0;
  for (uint32_t t = 0; t < nseq; ++t) {
    uint32_t L = (uint32_t)(offs[t + 1] - offs[t]);
    if (L > maxlen) maxlen = L;
    if (p->dbmask == 2) continue;                             /* LoadUDB loaddb.cpp:100-125: stored letters used as they are */
    if (p->dbmask) fastmask_impl(db->seqs + offs[t], L, p->dbmask == 3 ? (byte)(p->is_nucleo ? 'N' : 'X') : 0);
    else for (uint32_t i = 0; i < L; ++i) db->seqs[offs[t] + i] = (char)toupper((byte)db->seqs[offs[t] + i]);
  }
  db->maxlen = maxlen;
  /* alnparams.cpp:380-384 Init4 */
  float Open = p->is_nucleo ? -10.0f : -17.0f, Ext = -1.0f, TO = -0.5f, TE = -0.5f;
  db->ap.OpenA = db->ap.OpenB = Open;
  db->ap.LOpenA = db->ap.LOpenB = db->ap.ROpenA = db->ap.ROpenB = TO;
  db->ap.ExtA = db->ap.ExtB = Ext;
  db->ap.LExtA = db->ap.LExtB = db->ap.RExtA = db->ap.RExtB = TE;
  /* alnheuristics.cpp:26-62 */
  db->XDropGlobalHSP = p->xdrop_nw;
  db->BandRadius = (unsigned)p->band;
  db->MinGlobalHSPLength = (unsigned)p->minhsp;
  db->HSPw = (unsigned)p->hsp_word_len;
  float idf = p->id_set ? p->id : 0.5f;     /* oget_fltd(OPT_id, 0.5) */
  if (p->is_nucleo) {
    db->MinGlobalHSPFractId = idf > 0.75f ? idf : 0.75f;
    db->MinGlobalHSPScore = db->MinGlobalHSPFractId * db->MinGlobalHSPLength * p->match;
  } else {
    float MinDiag = 9e9f;
    static const char aa[] = "ACDEFGHIKLMNPQRSTVWY";
    for (int i = 0; i < 20; ++i) {
      float s = db->subst[(byte)aa[i]][(byte)aa[i]];
      if (s < MinDiag) MinDiag = s;
    }
    db->MinGlobalHSPFractId = idf > 0.5f ? idf : 0.5f;
    db->MinGlobalHSPScore = db->MinGlobalHSPFractId * MinDiag * db->MinGlobalHSPLength;
  }
  db->HSPWordCount = 1;
  for (unsigned i = 0; i < db->HSPw; ++i) db->HSPWordCount *= db->alpha;
  db->big = nseq > p->big;                  /* udbusortedsearcher.cpp:44 */
  int rc = build_index(db);
  if (rc != UGS_OK) { orc_db_destroy(db); return rc; }
  *out = db;
  return UGS_OK;
}

int orc_db_set_pair_keys(orc_db *db, const uint32_t *label_key, const uint32_t *size)
{
  free(db->t_key); free(db->t_size); db->t_key = db->t_size = NULL;
  if (label_key) { db->t_key = (uint32_t *)malloc((size_t)db->nseq * 4 + 4); memcpy(db->t_key, label_key, (size_t)db->nseq * 4); }
  if (size) { db->t_size = (uint32_t *)malloc((size_t)db->nseq * 4 + 4); memcpy(db->t_size, size, (size_t)db->nseq * 4); }
  return 0;
}
/* keys of the queries of the NEXT orc_search_batch (borrowed pointers) */
void orc_set_query_pair_keys(orc_db *db, const uint32_t *label_key, const uint32_t *size) { db->q_key = label_key; db->q_size = size; }

void orc_db_destroy(orc_db *db)
{
  if (!db) return;
  if (db->dyn_rows) { for (uint64_t s = 0; s < db->slots; ++s) free(db->dyn_rows[s]); free(db->dyn_rows); free(db->dyn_size); free(db->dyn_cap); }
  free(db->seqs); free(db->offs); free(db->row_off); free(db->postings); free(db->t_key); free(db->t_size); free(db);
}

const char *orc_db_masked(const orc_db *db) { return db->seqs; }
uint64_t orc_db_slots(const orc_db *db) { return db->slots; }
const uint64_t *orc_db_row_off(const orc_db *db) { return db->row_off; }
const uint32_t *orc_db_postings(const orc_db *db) { return db->postings; }
void orc_get_stats(const orc_db *db, orc_stats *st) { *st = db->stats; }

/* ------------------------------------------------------------------ per-thread workspace */

typedef struct {
  orc_db *db;
  /* ranking */
  uint32_t *qwords, *quniq; unsigned nqw, nquniq; uint32_t qcap;
  byte *wordfound;                 /* slots */
  uint32_t *U;                     /* nseq */
  uint32_t *top_t, *top_u, *order, *top_t2; /* nseq */
  uint32_t *cs_sizes, *cs_offsets; uint32_t cs_cap;
  unsigned ntop;                   /* candidates in final order: cand_t/cand_c */
  uint32_t *cand_t, *cand_c;
  /* hsp finder */
  uint32_t *wordsA, *wordsB; unsigned nwA, nwB, capA, capB;
  unsigned *wcountsA, *wposA;
  const byte *A, *B; unsigned LA, LB;
  HSP *hsps; unsigned nhsp, caphsp;
  HSP **chain; unsigned nchain;
  /* chainer scratch */
  unsigned *bp_pos, *bp_idx, *prev; byte *bp_islo; float *cscore; unsigned *list;
  /* dp */
  float *Mrow, *Drow; byte *TB; unsigned dpLA, dpLB; size_t tbcap, rowcap;
  char *path, *subpath; size_t pathcap;
  /* per-thread output */
  orc_stats st;
} Work;

static void *xrealloc(void *p, size_t n) { void *q = realloc(p, n ? n : 1); if (!q) abort(); return q; }

static Work *work_new_cap(orc_db *db, uint32_t nseq_cap)
{
  Work *w = (Work *)calloc(1, sizeof(Work));
  w->db = db;
  w->wordfound = (byte *)calloc(db->slots, 1);
  uint32_t n = nseq_cap ? nseq_cap : 1;
  w->U = (uint32_t *)calloc(n, 4);
  w->top_t = (uint32_t *)malloc(n * 4); w->top_u = (uint32_t *)malloc(n * 4);
  w->order = (uint32_t *)malloc(n * 4); w->top_t2 = (uint32_t *)malloc(n * 4);
  w->cand_t = (uint32_t *)malloc(n * 4); w->cand_c = (uint32_t *)malloc(n * 4);
  w->wcountsA = (unsigned *)malloc(db->HSPWordCount * sizeof(unsigned));
  w->wposA = (unsigned *)malloc((size_t)db->HSPWordCount * MAXREPS * sizeof(unsigned));
  return w;
}

static Work *work_new(orc_db *db) { return work_new_cap(db, db->nseq); }

static void work_free(Work *w)
{
  free(w->qwords); free(w->quniq); free(w->wordfound); free(w->U); free(w->top_t); free(w->top_u);
  free(w->order); free(w->top_t2); free(w->cs_sizes); free(w->cs_offsets); free(w->cand_t); free(w->cand_c);
  free(w->wordsA); free(w->wordsB); free(w->wcountsA); free(w->wposA); free(w->hsps); free(w->chain);
  free(w->bp_pos); free(w->bp_idx); free(w->prev); free(w->bp_islo); free(w->cscore); free(w->list);
  free(w->Mrow); free(w->Drow); free(w->TB); free(w->path); free(w->subpath); free(w);
}

/* ------------------------------------------------------------------ ranking */

/* udbsearcher.cpp:128-151 SetQueryWordsAllNoBadNoPattern + :161-194 SetQueryUniqueWords */
static void set_query_words(Work *w, const byte *q, unsigned L)
{
  orc_db *db = w->db;
  if (w->qcap < L + 1) {
    w->qcap = L + 1;
    w->qwords = (uint32_t *)xrealloc(w->qwords, w->qcap * 4);
    w->quniq = (uint32_t *)xrealloc(w->quniq, w->qcap * 4);
  }
  w->nqw = 0;
  if (L >= (unsigned)db->p.word_len)
    for (unsigned pos = 0; pos + db->p.word_len <= L; ++pos) {
      uint32_t word = seq_to_word(db, q + pos);
      if (word != BAD_WORD) w->qwords[w->nqw++] = word;
    }
  w->nquniq = 0;
  for (unsigned i = 0; i < w->nqw; ++i) {
    uint32_t word = w->qwords[i];
    if (!w->wordfound[word]) { w->quniq[w->nquniq++] = word; w->wordfound[word] = 1; }
  }
  for (unsigned i = 0; i < w->nqw; ++i) w->wordfound[w->qwords[i]] = 0;
}

/* wordparams.cpp:60-112 (table from CD-HIT, as the reference holds it) */
static const double MinWordFractAmino[50] = {
  0.00, 0.00, 0.00, 0.00, 0.01, 0.01, 0.01, 0.02, 0.02, 0.02, 0.03, 0.04, 0.04, 0.05, 0.06, 0.06, 0.08,
  0.08, 0.10, 0.10, 0.11, 0.14, 0.14, 0.14, 0.17, 0.17, 0.18, 0.20, 0.21, 0.21, 0.27, 0.28, 0.31, 0.34,
  0.36, 0.41, 0.43, 0.45, 0.48, 0.54, 0.55, 0.56, 0.64, 0.69, 0.73, 0.75, 0.80, 0.85, 0.90, 0.95};

/* wordparams.cpp:125-135,145-159,167-192 GetWordCountingParams -> Step (MinU unused by Big path) */
static unsigned word_step(const orc_db *db, unsigned Nu)
{
  double FractId = (double)db->p.id;   /* float widened to double, SURVEY A.2 */
  unsigned Thresh;
  if (db->p.is_nucleo) {
    double WordFract = 1 - (1 - FractId) * db->p.word_len;
    if (WordFract < 0.0) Thresh = 1;
    else {
      WordFract *= Nu;
      Thresh = WordFract < 1.0 ? 1 : (unsigned)WordFract;
    }
  } else {
    if (FractId < 0.5) Thresh = 0;
    else {
      unsigned i = (unsigned)((FractId - 0.5) * 100);
      if (i >= 50) i = 49;
      Thresh = (unsigned)(MinWordFractAmino[i] * Nu);
    }
  }
  if (db->p.stepwords == 0) return 1;
  unsigned Step = Thresh / db->p.stepwords;
  if (Step == 0) Step = 1;
  return Step;
}

static void cs_alloc(Work *w, unsigned N)
{
  if (w->cs_cap < N) {
    w->cs_cap = N + 64;
    w->cs_sizes = (uint32_t *)xrealloc(w->cs_sizes, w->cs_cap * 4);
    w->cs_offsets = (uint32_t *)xrealloc(w->cs_offsets, w->cs_cap * 4);
  }
}

/* countsort.cpp:6-108 CountSortOrderDesc */
static unsigned count_sort_order_desc(Work *w, const uint32_t *Values, unsigned ValueCount, uint32_t *Order)
{
  unsigned MaxValue = 0, NextValue = 0;
  for (unsigned i = 0; i < ValueCount; ++i) {
    unsigned v = Values[i];
    if (v > MaxValue) { NextValue = MaxValue; MaxValue = v; }
  }
  unsigned MinValue = NextValue / 2;
  unsigned N = MaxValue + 1;
  cs_alloc(w, N);
  memset(w->cs_sizes, 0, N * 4);
  for (unsigned i = 0; i < ValueCount; ++i) { unsigned v = Values[i]; if (v < MinValue) continue; ++w->cs_sizes[v]; }
  unsigned Offset = 0;
  for (int v = (int)MaxValue; v >= (int)MinValue; --v) { w->cs_offsets[v] = Offset; Offset += w->cs_sizes[v]; }
  for (unsigned i = 0; i < ValueCount; ++i) {
    unsigned v = Values[i];
    if (v < MinValue) continue;
    Order[w->cs_offsets[v]++] = i;
  }
  return w->cs_offsets[MinValue];
}

/* countsort.cpp:110-191 CountSortSubsetDesc */
static unsigned count_sort_subset_desc(Work *w, const uint32_t *Values, unsigned ValueCount,
                                       const uint32_t *Subset, uint32_t *Result)
{
  unsigned MaxValue = 0, NextValue = 0;
  for (unsigned i = 0; i < ValueCount; ++i) {
    unsigned v = Values[Subset[i]];
    if (v > MaxValue) { NextValue = MaxValue; MaxValue = v; }
  }
  unsigned MinValue = NextValue / 2;
  unsigned N = MaxValue + 1;
  cs_alloc(w, N);
  memset(w->cs_sizes, 0, N * 4);
  for (unsigned i = 0; i < ValueCount; ++i) { unsigned v = Values[Subset[i]]; if (v < MinValue) continue; ++w->cs_sizes[v]; }
  unsigned Offset = 0;
  for (int v = (int)MaxValue; v >= (int)MinValue; --v) { w->cs_offsets[v] = Offset; Offset += w->cs_sizes[v]; }
  for (unsigned i = 0; i < ValueCount; ++i) {
    unsigned k = Subset[i];
    unsigned v = Values[k];
    if (v < MinValue) continue;
    Result[w->cs_offsets[v]++] = k;
  }
  return w->cs_offsets[MinValue];
}

/* small path: udbusortedsearcher.cpp:109-120 SetTargetOrder = SetU_NonCoded(1) (:375-410),
 * SetTopBump(1, bump) (:230-267) or SetTopNoBump (:205-228), CountSortOrderDesc */
static void rank_small(Work *w, const byte *q, unsigned L)
{
  orc_db *db = w->db;
  set_query_words(w, q, L);
  w->ntop = 0;
  const unsigned SeqCount = db->nseq;
  if (SeqCount == 0) return;
  memset(w->U, 0, (size_t)SeqCount * 4);
  for (unsigned i = 0; i < w->nquniq; ++i) {
    uint32_t word = w->quniq[i];
    uint64_t size;
    const uint32_t *row = db_row(db, word, &size);
    w->st.postings += size;
    for (uint64_t j = 0; j < size; ++j) ++w->U[row[j]];
  }
  unsigned MinU = 1, TopCount = 0;
  if (db->p.bump_pct != 0) {
    double Bump = db->p.bump_pct / 100.0;
    unsigned MaxCount = 0;
    for (unsigned t = 0; t < SeqCount; ++t) {
      unsigned n = w->U[t];
      if (n >= MinU) {
        if (n > MaxCount) {
          unsigned NewMin = (unsigned)(n * Bump);
          if (NewMin > MinU && NewMin < MaxCount) MinU = NewMin;
          MaxCount = n;
        }
        w->top_u[TopCount] = n; w->top_t[TopCount] = t; ++TopCount;
      }
    }
  } else {
    for (unsigned t = 0; t < SeqCount; ++t) {
      unsigned n = w->U[t];
      if (n >= MinU) { w->top_u[TopCount] = n; w->top_t[TopCount] = t; ++TopCount; }
    }
  }
  unsigned K = count_sort_order_desc(w, w->top_u, TopCount, w->order);
  for (unsigned k = 0; k < K; ++k) {
    unsigned i = w->order[k];
    w->cand_t[k] = w->top_t[i]; w->cand_c[k] = w->top_u[i];
  }
  w->ntop = K;
}

/* Big path: udbusortedsearcherbig.cpp:31-135 scan part + CountSortSubsetDesc, then
 * udbusortedsearcher.cpp:65-84 OnQueryDoneImpl (clear touched counters) */
static void rank_big(Work *w, const byte *q, unsigned L)
{
  orc_db *db = w->db;
  w->ntop = 0;
  if (db->nseq == 0) return;
  set_query_words(w, q, L);
  unsigned Step = word_step(db, w->nquniq);
  unsigned TopCount = 0;
  for (unsigned i = 0; i < w->nquniq; i += Step) {
    uint32_t word = w->quniq[i];
    uint64_t size;
    const uint32_t *row = db_row(db, word, &size);
    w->st.postings += size;
    for (uint64_t j = 0; j < size; ++j) {
      uint32_t t = row[j];
      unsigned c = w->U[t];
      if (c == 0) w->top_t[TopCount++] = t;
      w->U[t] = c + 1;
    }
  }
  if (TopCount == 0) return;
  unsigned K = count_sort_subset_desc(w, w->U, TopCount, w->top_t, w->top_t2);
  for (unsigned k = 0; k < K; ++k) { w->cand_t[k] = w->top_t2[k]; w->cand_c[k] = w->U[w->top_t2[k]]; }
  w->ntop = K;
  for (unsigned i = 0; i < TopCount; ++i) w->U[w->top_t[i]] = 0;
}

/* ------------------------------------------------------------------ HSP finder */

/* hspfinder.cpp:226-270 SeqToWords: rolling word for EVERY position, invalid letter -> 0 */
static unsigned hf_seq_to_words(const orc_db *db, const byte *Seq, unsigned L, uint32_t *Words)
{
  const unsigned wl = db->HSPw, as = db->alpha;
  if (L < wl) return 0;
  const unsigned Hi = db->HSPWordCount / as;
  uint32_t Word = 0;
  const byte *Front = Seq, *Back = Seq;
  for (unsigned i = 0; i + 1 < wl; ++i) {
    unsigned Letter = db->c2l[*Front++];
    if (Letter >= as) Letter = 0;
    Word = Word * as + Letter;
  }
  for (unsigned i = wl - 1; i < L; ++i) {
    unsigned Letter = db->c2l[*Front++];
    if (Letter >= as) Letter = 0;
    Word = Word * as + Letter;
    *Words++ = Word;
    Letter = db->c2l[*Back++];
    if (Letter >= as) Letter = 0;
    Word -= Letter * Hi;
  }
  return L - wl + 1;
}

/* hspfinder.cpp:304-323 SetA */
static void hf_set_a(Work *w, const byte *A, unsigned LA)
{
  orc_db *db = w->db;
  if (w->capA < LA + 1) { w->capA = LA + 512; w->wordsA = (uint32_t *)xrealloc(w->wordsA, w->capA * 4); }
  memset(w->wcountsA, 0, db->HSPWordCount * sizeof(unsigned));
  w->A = A; w->LA = LA;
  w->nwA = hf_seq_to_words(db, A, LA, w->wordsA);
  for (unsigned PosA = 0; PosA < w->nwA; ++PosA) {
    unsigned Word = w->wordsA[PosA];
    unsigned n = w->wcountsA[Word];
    if (n == MAXREPS) continue;
    w->wposA[Word * MAXREPS + n] = PosA;
    ++w->wcountsA[Word];
  }
}

/* hspfinder.cpp:325-331 SetB */
static void hf_set_b(Work *w, const byte *B, unsigned LB)
{
  if (w->capB < LB + 1) { w->capB = LB + 32; w->wordsB = (uint32_t *)xrealloc(w->wordsB, w->capB * 4); }
  w->B = B; w->LB = LB;
  w->nwB = hf_seq_to_words(w->db, B, LB, w->wordsB);
}

/* hspfinder.cpp:594-636 IsGlobalHSP */
static int is_global_hsp(unsigned ALo, unsigned BLo, unsigned Length, unsigned LA, unsigned LB)
{
  (void)Length;
  if (LA <= LB) {
    unsigned MaxGap = LA / 4 + 1;
    if (ALo > BLo && ALo - BLo > MaxGap) return 0;
    unsigned AR = LA - ALo, BR = LB - BLo;
    if (AR > BR && AR - BR > MaxGap) return 0;
  } else {
    unsigned MaxGap = LB / 4 + 1;
    if (BLo > ALo && BLo - ALo > MaxGap) return 0;
    unsigned AR = LA - ALo, BR = LB - BLo;
    if (BR > AR && BR - AR > MaxGap) return 0;
  }
  return 1;
}

/* ungappedblast.cpp:8-211 UngappedBlast(X, StaggerOk=false, MinLength, MinScore) */
static void ungapped_blast(Work *w, float X, unsigned MinLength, float MinScore)
{
  orc_db *db = w->db;
  w->nhsp = 0;
  ++w->st.ungapped_calls;
  const unsigned wl = db->HSPw;
  if (w->LB < 2 * wl) return;
  const byte *A = w->A, *B = w->B;
  const unsigned LA = w->LA, LB = w->LB;
  float (*Mx)[256] = db->subst;
  unsigned BPos = 0;
  for (;;) {
    if (BPos >= w->nwB) break;
    unsigned Word = w->wordsB[BPos];
    unsigned NA = w->wcountsA[Word];
    if (NA == 0) { ++BPos; continue; }
    int found = 0;
    for (unsigned i = 0; i < NA; ++i) {
      unsigned APos = w->wposA[Word * MAXREPS + i];
      unsigned Diag = (LA + BPos) - APos;
      unsigned BPos2 = BPos + wl - 1, APos2 = APos + wl - 1;
      if (APos2 >= LA || BPos2 >= LB) continue;
      float Score = 0;
      for (unsigned j = 0; j < wl; ++j) Score += Mx[A[APos + j]][B[BPos + j]];
      float BestScore = Score;
      unsigned BestBPos2 = BPos2;
      for (;;) {                                   /* extend right */
        ++BPos2; if (BPos2 >= LB) break;
        ++APos2; if (APos2 >= LA) break;
        Score += Mx[A[APos2]][B[BPos2]];
        if (Score > BestScore) { BestScore = Score; BestBPos2 = BPos2; }
        else if (BestScore - Score > X) break;
      }
      unsigned APos1 = APos, BPos1 = BPos, BestBPos1 = BPos1;   /* extend left */
      Score = BestScore;
      for (;;) {
        if (BPos1 == 0 || APos1 == 0) break;
        --BPos1; --APos1;
        Score += Mx[A[APos1]][B[BPos1]];
        if (Score > BestScore) { BestScore = Score; BestBPos1 = BPos1; }
        else if (BestScore - Score > X) break;
      }
      unsigned Blo = BestBPos1, Bhi = BestBPos2;
      unsigned Length = Bhi - Blo + 1;
      unsigned Alo = (LA + BestBPos1) - Diag;
      int Ok = (Length >= MinLength && BestScore >= MinScore);
      Ok = Ok && is_global_hsp(Alo, Blo, Length, LA, LB);
      if (Ok) {
        if (w->nhsp + 1 > w->caphsp) {
          w->caphsp = w->caphsp * 2 + 64;
          w->hsps = (HSP *)xrealloc(w->hsps, w->caphsp * sizeof(HSP));
        }
        HSP *h = &w->hsps[w->nhsp++];
        h->Loi = Alo; h->Loj = Blo; h->Len = Length; h->Score = BestScore;
        BPos = Bhi + 1;
        found = 1;
        break;
      }
    }
    if (!found) ++BPos;
  }
}

/* stable merge sort of break points (glibc qsort with enough memory is a stable
 * merge sort; comparator chainer.cpp:219-238: Pos asc, Lo before Hi, else equal) */
static int bp_less(unsigned posa, int loa, unsigned posb, int lob)
{
  if (posa != posb) return posa < posb;
  if (loa != lob) return loa && !lob;
  return 0;
}

/* chainer.cpp:352-500 Chain (+ :322-350 FindBestChainLT, :251-270 SetBPs); the
 * "delete enclosed chains" branch compares a value with itself and never fires (:447-448) */
static void chain_hsps(Work *w)
{
  const unsigned n = w->nhsp;
  w->nchain = 0;
  if (n == 0) return;
  w->chain = (HSP **)xrealloc(w->chain, n * sizeof(HSP *));
  w->bp_pos = (unsigned *)xrealloc(w->bp_pos, 2 * n * sizeof(unsigned));
  w->bp_idx = (unsigned *)xrealloc(w->bp_idx, 2 * n * sizeof(unsigned));
  w->bp_islo = (byte *)xrealloc(w->bp_islo, 2 * n);
  w->prev = (unsigned *)xrealloc(w->prev, n * sizeof(unsigned));
  w->cscore = (float *)xrealloc(w->cscore, n * sizeof(float));
  w->list = (unsigned *)xrealloc(w->list, n * sizeof(unsigned));
  for (unsigned i = 0; i < n; ++i) {
    w->bp_pos[2 * i] = w->hsps[i].Loi; w->bp_islo[2 * i] = 1; w->bp_idx[2 * i] = i;
    w->bp_pos[2 * i + 1] = w->hsps[i].Loi + w->hsps[i].Len - 1; w->bp_islo[2 * i + 1] = 0; w->bp_idx[2 * i + 1] = i;
  }
  /* stable insertion sort == stable merge sort result */
  for (unsigned i = 1; i < 2 * n; ++i) {
    unsigned p = w->bp_pos[i], ix = w->bp_idx[i]; byte lo = w->bp_islo[i];
    unsigned j = i;
    while (j > 0 && bp_less(p, lo, w->bp_pos[j - 1], w->bp_islo[j - 1])) {
      w->bp_pos[j] = w->bp_pos[j - 1]; w->bp_idx[j] = w->bp_idx[j - 1]; w->bp_islo[j] = w->bp_islo[j - 1];
      --j;
    }
    w->bp_pos[j] = p; w->bp_idx[j] = ix; w->bp_islo[j] = lo;
  }
  for (unsigned i = 0; i < n; ++i) w->prev[i] = UINT_MAX;
  unsigned nlist = 0;
  for (unsigned b = 0; b < 2 * n; ++b) {
    unsigned hi = w->bp_idx[b];
    const HSP *h = &w->hsps[hi];
    if (!w->bp_islo[b]) continue;
    unsigned Ahi = h->Loi, Bhi = h->Loj;
    float BestScore = -9e9f;
    unsigned BestChain = UINT_MAX;
    for (unsigned k = 0; k < nlist; ++k) {
      unsigned ci = w->list[k];
      const HSP *c = &w->hsps[ci];
      unsigned cAhi = c->Loi + c->Len - 1, cBhi = c->Loj + c->Len - 1;
      float cs = w->cscore[ci];
      if (cAhi < Ahi && cBhi < Bhi && (BestChain == UINT_MAX || cs > BestScore)) { BestChain = ci; BestScore = cs; }
    }
    w->list[nlist++] = hi;
    w->prev[hi] = BestChain;
    w->cscore[hi] = BestChain == UINT_MAX ? h->Score : w->cscore[BestChain] + h->Score;
  }
  unsigned Opt = 0;
  float OptScore = w->cscore[0];
  for (unsigned i = 1; i < n; ++i)
    if (w->cscore[i] > OptScore) { Opt = i; OptScore = w->cscore[i]; }
  unsigned len = 0;
  for (unsigned i = Opt; i != UINT_MAX; i = w->prev[i]) ++len;
  unsigned k = 1;
  for (unsigned i = Opt; i != UINT_MAX; i = w->prev[i]) w->chain[len - k++] = &w->hsps[i];
  w->nchain = len;
}

/* hsp.h:102-126 IsStaggered */
static int hsp_is_staggered(const HSP *h, unsigned LA, unsigned LB)
{
  int Hii = (int)(h->Loi + h->Len - 1), Hij = (int)(h->Loj + h->Len - 1);
  int TermGapLeftA = (int)h->Loi - (int)h->Loj;
  int TermGapLeftB = (int)h->Loj - (int)h->Loi;
  int TermGapRightA = (int)LA - Hii - 1 - ((int)LB - Hij - 1);
  int TermGapRightB = (int)LB - Hij - 1 - ((int)LA - Hii - 1);
  if (TermGapLeftA < 0) TermGapLeftA = 0;
  if (TermGapLeftB < 0) TermGapLeftB = 0;
  if (TermGapRightB < 0) TermGapRightB = 0;     /* (TermGapRightA is NOT clamped in the reference) */
  int GapA = TermGapLeftA + TermGapRightA;
  int GapB = TermGapLeftB + TermGapRightB;
  if (GapA == 0 || GapB == 0) return 0;
  double r = (LA < LB ? (double)GapA / LA : (double)GapB / LB);
  return r > 0.5;
}

/* getglobalhsps.cpp:9-61 GetGlobalHSPs (+ hspfinder.cpp:537-553 Chain, :561-579 GetHSPIdCount) */
static unsigned get_global_hsps(Work *w, unsigned MinLength, float *HSPFractId)
{
  orc_db *db = w->db;
  ungapped_blast(w, db->XDropGlobalHSP, MinLength, db->MinGlobalHSPScore);
  chain_hsps(w);
  for (unsigned i = 0; i < w->nchain; ++i)
    if (hsp_is_staggered(w->chain[i], w->LA, w->LB)) { w->nchain = 0; break; }
  unsigned TotalLength = 0, TotalSame = 0;
  for (unsigned i = 0; i < w->nchain; ++i) {
    const HSP *h = w->chain[i];
    TotalLength += h->Len;
    for (unsigned k = 0; k < h->Len; ++k)
      if (db->match[w->A[h->Loi + k]][w->B[h->Loj + k]]) ++TotalSame;
  }
  *HSPFractId = TotalLength == 0 ? 0.0f : (float)TotalSame / (float)TotalLength;
  return w->nchain;
}

/* ------------------------------------------------------------------ banded Viterbi */

static void dp_alloc(Work *w, unsigned LA, unsigned LB)
{
  size_t need_row = (size_t)LB + 8;
  if (w->rowcap < need_row) {
    w->rowcap = need_row + 256;
    w->Mrow = (float *)xrealloc(w->Mrow, w->rowcap * sizeof(float));
    w->Drow = (float *)xrealloc(w->Drow, w->rowcap * sizeof(float));
  }
  size_t need_tb = ((size_t)LA + 1) * ((size_t)LB + 1);
  if (w->tbcap < need_tb) { w->tbcap = need_tb + 1024; w->TB = (byte *)xrealloc(w->TB, w->tbcap); }
}

static void path_alloc(Work *w, size_t n)
{
  if (w->pathcap < n + 16) {
    w->pathcap = n + 1024;
    w->path = (char *)xrealloc(w->path, w->pathcap);
    w->subpath = (char *)xrealloc(w->subpath, w->pathcap);
  }
}

/* diagbox.h:150-171 GetRange_j */
static void get_range_j(unsigned LA, unsigned LB, unsigned dlo, unsigned dhi, unsigned i,
                        unsigned *Startj, unsigned *Endj)
{
  unsigned s = (dlo + i >= LA) ? dlo + i - LA : 0;
  if (s >= LB) s = LB - 1;
  unsigned e = (dhi + i + 1 >= LA) ? dhi + i + 1 - LA : 0;
  if (e > LB) e = LB;
  *Startj = s; *Endj = e;
}

/* viterbifastbandmem.cpp:12-230 ViterbiFastBandMem + tracebackbitmem.cpp:8-73; writes the
 * path (M/D/I text, NUL-terminated) into out */
static float viterbi_band(Work *w, const byte *A, unsigned LA, const byte *B, unsigned LB,
                          unsigned DiagLo, unsigned DiagHi, const AlnPen *AP, char *out)
{
  orc_db *db = w->db;
  dp_alloc(w, LA, LB);
  float (*Mx)[256] = db->subst;
  float OpenA = AP->LOpenA, ExtA = AP->LExtA;
  float *Mrow = w->Mrow + 1, *Drow = w->Drow + 1;   /* Mrow[-1] is addressable */
  byte *TB = w->TB;
  const size_t TBS = (size_t)LB + 1;
  Mrow[-1] = MINUS_INFINITY;
  for (unsigned j = 0; j <= LB; ++j) { Mrow[j] = MINUS_INFINITY; Drow[j] = MINUS_INFINITY; }
  uint64_t cells = 0;
  for (unsigned i = 0; i < LA; ++i) {
    unsigned Startj, Endj;
    get_range_j(LA, LB, DiagLo, DiagHi, i, &Startj, &Endj);
    if (Endj == 0) continue;
    float OpenB = Startj == 0 ? AP->LOpenB : AP->OpenB;
    float ExtB = Startj == 0 ? AP->LExtB : AP->ExtB;
    const float *MxRow = Mx[A[i]];
    float I0 = MINUS_INFINITY;
    float M0;
    if (i == 0) M0 = 0;
    else M0 = (Startj == 0) ? MINUS_INFINITY : Mrow[(int)Startj - 1];
    byte *TBrow = TB + (size_t)i * TBS;
    if (Startj > 0) TBrow[(int)Startj - 1] = TB_IM;
    cells += Endj - Startj;
    for (unsigned j = Startj; j < Endj; ++j) {
      byte b = B[j];
      byte TraceBits = 0;
      float SavedM0 = M0;
      {
        float xM = M0;
        if (Drow[j] > xM) { xM = Drow[j]; TraceBits = TB_DM; }
        if (I0 > xM) { xM = I0; TraceBits = TB_IM; }
        M0 = Mrow[j];
        Mrow[j] = xM + MxRow[b];
      }
      {
        float md = SavedM0 + OpenB;
        Drow[j] += ExtB;
        if (md >= Drow[j]) { Drow[j] = md; TraceBits |= TB_MD; }
      }
      {
        float mi = SavedM0 + OpenA;
        I0 += ExtA;
        if (mi >= I0) { I0 = mi; TraceBits |= TB_MI; }
      }
      OpenB = AP->OpenB; ExtB = AP->ExtB;
      TBrow[j] = TraceBits;
    }
    {
      TBrow[LB] = 0;
      float md = M0 + AP->ROpenB;
      Drow[LB] += AP->RExtB;
      if (md >= Drow[LB]) { Drow[LB] = md; TBrow[LB] = TB_MD; }
    }
    M0 = MINUS_INFINITY;
    OpenA = AP->OpenA; ExtA = AP->ExtA;
  }
  unsigned Startj, Endj;
  get_range_j(LA, LB, DiagLo, DiagHi, LA - 1, &Startj, &Endj);
  byte *TBrow = TB + (size_t)LA * TBS;
  float I1 = MINUS_INFINITY;
  Mrow[(int)Startj - 1] = MINUS_INFINITY;
  cells += LB;   /* last-row I sweep, SURVEY 8d */
  for (unsigned j = Startj; j < Endj; ++j) {
    TBrow[j] = 0;
    float mi = Mrow[(int)j - 1] + AP->ROpenA;
    I1 += AP->RExtA;
    if (mi > I1) { I1 = mi; TBrow[j] = TB_MI; }
  }
  float FinalM = Mrow[LB - 1], FinalD = Drow[LB], FinalI = I1;
  float Score = FinalM; char State = 'M';
  if (FinalD > Score) { Score = FinalD; State = 'D'; }
  if (FinalI > Score) { Score = FinalI; State = 'I'; }
  w->st.dp_cells += cells; ++w->st.dp_calls;

  /* tracebackbitmem.cpp:8-73 */
  size_t i = LA, j = LB, n = 0;
  for (;;) {
    if (i == 0 && j == 0) break;
    out[n++] = State;
    byte t;
    switch (State) {
    case 'M':
      t = TB[(i - 1) * TBS + (j - 1)];
      if (t & TB_DM) State = 'D'; else if (t & TB_IM) State = 'I'; else State = 'M';
      --i; --j; break;
    case 'D':
      t = TB[(i - 1) * TBS + j];
      State = (t & TB_MD) ? 'M' : 'D';
      --i; break;
    default:
      t = TB[i * TBS + (j - 1)];
      State = (t & TB_MI) ? 'M' : 'I';
      --j; break;
    }
  }
  for (size_t k = 0; k < n / 2; ++k) { char c = out[k]; out[k] = out[n - 1 - k]; out[n - 1 - k] = c; }
  out[n] = 0;
  return Score;
}

/* viterbifastbandmem.cpp:232-253 ViterbiFastMainDiagMem */
static float viterbi_main_diag(Work *w, const byte *A, unsigned LA, const byte *B, unsigned LB,
                               unsigned BandRadius, const AlnPen *AP, char *out)
{
  /* -band 0: GlobalAlignBandMem / AlignHSPMem call ViterbiFastMem instead (globalalignmem.cpp:105-108,118-119) = every diagonal */
  if (BandRadius == 0) return viterbi_band(w, A, LA, B, LB, 1, LA + LB - 1, AP, out);
  unsigned DiagLo = LA < LB ? LA : LB;
  unsigned DiagHi = LA > LB ? LA : LB;
  if (DiagLo > BandRadius) DiagLo -= BandRadius; else DiagLo = 1;
  DiagHi += BandRadius;
  unsigned MaxDiag = LA + LB - 1;
  if (DiagHi > MaxDiag) DiagHi = MaxDiag;
  return viterbi_band(w, A, LA, B, LB, DiagLo, DiagHi, AP, out);
}

/* globalalignmem.cpp:70-112 AlignHSPMem with alnparams.cpp:100-152 AlnParams::Init */
static void align_hole(Work *w, unsigned Loi, unsigned Loj, unsigned Leni, unsigned Lenj, char *out)
{
  orc_db *db = w->db;
  out[0] = 0;
  if (Leni == 0) { if (Lenj > 0) { memset(out, 'I', Lenj); out[Lenj] = 0; } return; }
  if (Lenj == 0) { memset(out, 'D', Leni); out[Leni] = 0; return; }
  const AlnPen *AP = &db->ap;
  AlnPen L = *AP;
  int LeftA = (Loi == 0), LeftB = (Loj == 0);
  int RightA = (Loi + Leni == w->LA), RightB = (Loj + Lenj == w->LB);
  L.LOpenA = LeftA ? AP->LOpenA : AP->OpenA;   L.LExtA = LeftA ? AP->LExtA : AP->ExtA;
  L.LOpenB = LeftB ? AP->LOpenB : AP->OpenB;   L.LExtB = LeftB ? AP->LExtB : AP->ExtB;
  L.ROpenA = RightA ? AP->ROpenA : AP->OpenA;  L.RExtA = RightA ? AP->RExtA : AP->ExtA;
  L.ROpenB = RightB ? AP->ROpenB : AP->OpenB;  L.RExtB = RightB ? AP->RExtB : AP->ExtB;
  viterbi_main_diag(w, w->A + Loi, Leni, w->B + Loj, Lenj, db->BandRadius, &L, out);
}

/* globalalignmem.cpp:129-236 GlobalAlign_AllOpts (FullDPAlways=false, FailIfNoHSPs=true).
 * SetA must have been called for the query; returns 1 and w->path on success. */
static int global_align(Work *w, const byte *B, unsigned LB, float *HSPFractIdOut)
{
  orc_db *db = w->db;
  const unsigned LA = w->LA;
  hf_set_b(w, B, LB);
  path_alloc(w, (size_t)LA + LB + 2);
  unsigned MinHSPLength = db->MinGlobalHSPLength == 0 ? 32 : db->MinGlobalHSPLength;
  if (MinHSPLength > LA / 4) MinHSPLength = LA / 4;
  if (MinHSPLength < 16) MinHSPLength = 16;
  float HSPFractId = -1.0f;
  const int FailIfNoHSPs = !(db->p.align_flags & UGS_A_GAFORCE);            /* globalaligner.cpp:9-12 */
  if (db->p.align_flags & UGS_A_FULLDP) {                                     /* globalalignmem.cpp:148-152: ViterbiFastMem = every diagonal */
    if (HSPFractIdOut) *HSPFractIdOut = HSPFractId;
    viterbi_band(w, w->A, LA, B, LB, 1, LA + LB - 1, &db->ap, w->path);
    return 1;
  }
  unsigned HSPCount = get_global_hsps(w, MinHSPLength, &HSPFractId);
  if (HSPFractIdOut) *HSPFractIdOut = HSPFractId;
  if (HSPFractId < db->MinGlobalHSPFractId && FailIfNoHSPs) return 0;
  if (HSPCount == 0) {
    if (db->MinGlobalHSPLength > 0 && LA > 64 && FailIfNoHSPs) return 0;
    viterbi_main_diag(w, w->A, LA, B, LB, db->BandRadius, &db->ap, w->path);
    return 1;
  }
  char *p = w->path;
  for (unsigned i = 0; i < HSPCount; ++i) {
    const HSP *Prev = i == 0 ? NULL : w->chain[i - 1];
    const HSP *H = w->chain[i];
    unsigned hLoi, hLoj, hLeni, hLenj;       /* globalalignmem.cpp:25-68 GetHole */
    if (Prev) {
      unsigned pHii = Prev->Loi + Prev->Len - 1, pHij = Prev->Loj + Prev->Len - 1;
      hLoi = pHii + 1; hLoj = pHij + 1; hLeni = H->Loi - pHii - 1; hLenj = H->Loj - pHij - 1;
    } else { hLoi = 0; hLoj = 0; hLeni = H->Loi; hLenj = H->Loj; }
    align_hole(w, hLoi, hLoj, hLeni, hLenj, w->subpath);
    size_t n = strlen(w->subpath);
    memcpy(p, w->subpath, n); p += n;
    memset(p, 'M', H->Len); p += H->Len;
  }
  {
    const HSP *Last = w->chain[HSPCount - 1];
    unsigned hLoi = Last->Loi + Last->Len, hLoj = Last->Loj + Last->Len;
    align_hole(w, hLoi, hLoj, LA - hLoi, LB - hLoj, w->subpath);
    size_t n = strlen(w->subpath);
    memcpy(p, w->subpath, n); p += n;
  }
  *p = 0;
  return 1;
}

/* ------------------------------------------------------------------ hits */

/* arscorer.cpp:201-296 FillLo + :554-569 GetGapOpenCount; path covers both whole sequences */
static void fill_hit(const orc_db *db, const char *Path, const byte *Q, unsigned QL, const byte *T,
                     unsigned TL, ugs_hit *h)
{
  unsigned FirstM = UINT_MAX, LastM = UINT_MAX, Col = 0;
  for (; Path[Col]; ++Col) if (Path[Col] == 'M') { if (FirstM == UINT_MAX) FirstM = Col; LastM = Col; }
  unsigned ColCount = Col;
  unsigned FirstQ = 0, FirstT = 0;
  for (unsigned c = 0; c < FirstM && c < ColCount; ++c) {
    if (Path[c] == 'M' || Path[c] == 'D') ++FirstQ;
    if (Path[c] == 'M' || Path[c] == 'I') ++FirstT;
  }
  unsigned QPos = FirstQ, TPos = FirstT, Ids = 0, Mism = 0, IntGaps = 0, Opens = 0;
  char Lastc = 'M';
  for (unsigned c = FirstM; c <= LastM && FirstM != UINT_MAX; ++c) {
    char ch = Path[c];
    if (ch == 'M') {
      if (db->match[Q[QPos]][T[TPos]]) ++Ids; else ++Mism;
      ++QPos; ++TPos;
    } else if (ch == 'D') { if (c > FirstM) ++IntGaps; ++QPos; }
    else { if (c > FirstM) ++IntGaps; ++TPos; }
    if (ch != 'M' && Lastc == 'M') ++Opens;
    Lastc = ch;
  }
  h->ids = Ids; h->mism = Mism; h->gaps_int = IntGaps; h->opens = Opens;
  h->aln_len = LastM - FirstM + 1;
  h->qlo = FirstQ; h->tlo = FirstT; h->qhi = QPos - 1; h->thi = TPos - 1;
  h->ql = QL; h->tl = TL; h->cols = ColCount;
}

/* sort.h:85-117 QuickSortOrderRecurse<float, Desc=true> */
static void qs_order_desc(const float *V, int left, int right, unsigned *Order)
{
  int i = left, j = right;
  float pivot = V[Order[(left + right) / 2]];
  while (i <= j) {
    while (V[Order[i]] > pivot) i++;
    while (V[Order[j]] < pivot) j--;
    if (i <= j) { unsigned t = Order[i]; Order[i] = Order[j]; Order[j] = t; i++; j--; }
  }
  if (left < j) qs_order_desc(V, left, j, Order);
  if (i < right) qs_order_desc(V, i, right, Order);
}

typedef struct {
  ugs_hit *hits; unsigned nhits, cap;
  uint32_t *cig; uint64_t ncig, cigcap;
} HitBuf;

static void hb_push(HitBuf *hb, const ugs_hit *h, const char *path)
{
  if (hb->nhits + 1 > hb->cap) { hb->cap = hb->cap * 2 + 64; hb->hits = (ugs_hit *)xrealloc(hb->hits, hb->cap * sizeof(ugs_hit)); }
  ugs_hit *o = &hb->hits[hb->nhits++];
  *o = *h;
  o->cigar_off = hb->ncig;
  uint32_t runs = 0;
  for (const char *p = path; *p;) {
    char c = *p; uint32_t n = 0;
    while (*p == c) { ++p; ++n; }
    if (hb->ncig + 1 > hb->cigcap) { hb->cigcap = hb->cigcap * 2 + 256; hb->cig = (uint32_t *)xrealloc(hb->cig, hb->cigcap * 4); }
    hb->cig[hb->ncig++] = (n << 2) | (c == 'M' ? 0u : c == 'D' ? 1u : 2u);
    ++runs;
  }
  o->cigar_len = runs;
}

/* searcher.cpp:122-161 Search for one strand: SetQueryImpl/Aligner::SetQuery/SearchImpl with the
 * candidate loop (udbusortedsearcher.cpp:138-151 / udbusortedsearcherbig.cpp:113-134), OnAR
 * (searcher.cpp:52-61), Accepter::IsAcceptLo (accepter.cpp:27-94, default filters), Terminator
 * (terminator.cpp:64-100) */
/* Accepter::IsAcceptLo accepter.cpp:24-91 on the FillLo statistics of a global hit (coverages: arscorer.cpp:122-154) */
/* Accepter::RejectPair accepter.cpp:140-197 (Global accepter).  q is the strand's sequence as searched. */
static int reject_pair(const orc_db *db, uint32_t qindex, const byte *q, unsigned QL, uint32_t t, const byte *T, unsigned TL)
{
  const ugs_params *p = &db->p;
  const unsigned m = p->pair_mask;
  if (!m) return 0;
  if (m & (UGS_P_SELF | UGS_P_NOTSELF)) {
    int same = db->q_key && db->t_key && db->q_key[qindex] == db->t_key[t];
    if ((m & UGS_P_SELF) && same) return 1;
    if ((m & UGS_P_NOTSELF) && !same) return 1;
  }
  if ((m & UGS_P_SELFID) && TL == QL && memcmp(q, T, QL) == 0) return 1;
  if (m & UGS_P_MIN_SIZERATIO) {
    unsigned QS = db->q_size ? db->q_size[qindex] : UINT_MAX, TS = db->t_size ? db->t_size[t] : UINT_MAX;
    double Ratio = (double)TS / (double)QS;
    if (Ratio < (double)p->min_sizeratio) return 1;
  }
  if (m & (UGS_P_MINQT | UGS_P_MAXQT | UGS_P_MINSL | UGS_P_MAXSL)) {
    double qq = (double)QL, tt = (double)TL, ss = (double)(QL < TL ? QL : TL), ll = (double)(QL > TL ? QL : TL);
    double qt = qq / tt, sl = ss / ll;
    if ((m & UGS_P_MINQT) && qt < (double)p->minqt) return 1;
    if ((m & UGS_P_MAXQT) && qt > (double)p->maxqt) return 1;
    if ((m & UGS_P_MINSL) && sl < (double)p->minsl) return 1;
    if ((m & UGS_P_MAXSL) && sl > (double)p->maxsl) return 1;
  }
  return 0;
}

static int is_accept_lo(const ugs_params *p, const ugs_hit *h)
{
  const unsigned m = p->filter_mask;
  if (p->id_set) {
    double FractId = h->aln_len == 0 ? 0.0 : (double)h->ids / (double)h->aln_len;
    if (FractId < p->id_accept) return 0;
    if ((m & UGS_F_MAXID) && FractId > (double)p->maxid) return 0;
  }
  if ((m & UGS_F_MINCOLS) && h->aln_len < p->mincols) return 0;
  if ((m & UGS_F_MAXGAPS) && h->gaps_int > p->maxgaps) return 0;
  if (m & (UGS_F_QUERY_COV | UGS_F_MAX_QUERY_COV)) {
    unsigned n = h->qhi - h->qlo + 1;
    double Cov = (double)n / h->ql;
    if ((m & UGS_F_QUERY_COV) && Cov < (double)p->query_cov) return 0;
    if ((m & UGS_F_MAX_QUERY_COV) && Cov > (double)p->max_query_cov) return 0;
  }
  if (m & (UGS_F_TARGET_COV | UGS_F_MAX_TARGET_COV)) {
    unsigned MCount = h->ids + h->mism;
    double Cov = (double)MCount / (double)h->tl;
    if ((m & UGS_F_TARGET_COV) && Cov < (double)p->target_cov) return 0;
    if ((m & UGS_F_MAX_TARGET_COV) && Cov > (double)p->max_target_cov) return 0;
  }
  if ((m & UGS_F_MAXDIFFS) && h->mism + h->gaps_int > p->maxdiffs) return 0;
  if ((m & UGS_F_MINDIFFS) && h->mism + h->gaps_int < p->mindiffs) return 0;
  return 1;
}


/* ================================================================== usearch_local driver
 * Searcher::Align non-global branch (searcher.cpp:28-50) -> LocalAligner2::AlignMulti (localmulti.cpp:9-118)
 * -> LocalAligner::AlignPos (localaligner.cpp:101-222) -> XDropAlignMem, gated by Karlin-Altschul
 * statistics (estats.cpp:25-99). */
int orc_xdrop_job(const ugs_xdrop_params *p, const char *a, uint32_t la, const char *b, uint32_t lb,
                  const ugs_xdrop_job *job, ugs_xdrop_hsp *hsp, char *path, uint64_t *cells);

typedef struct { double GappedLambda, UngappedLambda, GappedK, UngappedK, LogGappedK, LogUngappedK, DBSize, MaxEvalue; } EStats;

static void es_init(EStats *es, const ugs_params *p)          /* estats.cpp:25-58, makedbsearcher.cpp:89-96 */
{
  if (p->is_nucleo) { es->GappedLambda = 1.280; es->UngappedLambda = 1.330; es->GappedK = 0.460; es->UngappedK = 0.621; }
  else { es->GappedLambda = 0.267; es->UngappedLambda = 0.311; es->GappedK = 0.0410; es->UngappedK = 0.128; }
  es->LogGappedK = log(es->GappedK); es->LogUngappedK = log(es->UngappedK);
  es->DBSize = (double)p->ka_dbsize;                          /* float DBSize, widened */
  es->MaxEvalue = (double)p->evalue;                          /* (float) oget_flt(OPT_evalue) */
}
/* The reference is built with -O3 -ffast-math (its Makefile:11-14), and the e-values it PRINTS depend on that: gcc
 * folds (x/Log2)*Log2, turns /Log2 into a multiplication by 1/ln 2 and NM/pow(2,b) into exp2(-b)*DBSize*QL, which
 * keeps e-values below 1e-292 representable (plain pow(2, bits > 1024) would overflow to E = 0).  The three functions
 * below are the operations of the compiled code (objdump of estats.o, gcc 11.4), not of the source text. */
static double es_min_ungapped(const EStats *es, unsigned QL)  /* estats.cpp:63-70 */
{
  return ((log((double)QL * es->DBSize) + es->LogUngappedK) - log(es->MaxEvalue)) / es->UngappedLambda;
}
static double es_bits(const EStats *es, double Raw, int Gapped)      /* estats.cpp:78-84 */
{
  double Lambda = Gapped ? es->GappedLambda : es->UngappedLambda, LogK = Gapped ? es->LogGappedK : es->LogUngappedK;
  return (Raw * Lambda - LogK) * 1.4426950408889634;          /* 0x3ff71547652b82fe */
}
static double es_evalue(const EStats *es, double Raw, unsigned QL, int Gapped)   /* estats.cpp:72-76,87-96 */
{
  return (exp2(-es_bits(es, Raw, Gapped)) * es->DBSize) * (double)QL;
}
void orc_local_evalue(const ugs_params *p, double raw, uint32_t ql, double *evalue, double *bits)
{
  EStats es; es_init(&es, p);
  if (bits) *bits = es_bits(&es, raw, 1);
  if (evalue) *evalue = es_evalue(&es, raw, ql, 1);
}

typedef struct {
  unsigned W, Alpha, Dict, AlphaHi;
  uint32_t *QueryWords, *QueryPosVec, *Counts, *Counts2, *Base; unsigned nwords, qcap;
  uint32_t *TargetWords; unsigned tcap;
  EStats es; float MinUngapped;
  ugs_xdrop_params xp;
  /* ARs of the current target */
  struct LAR { unsigned Loi, Loj, Leni, Lenj; float Score; char *Path; } *ars; unsigned nars, arcap;
  char *path; size_t pathcap;
} LocalWork;

static unsigned long g_local_rescore_diffs = 0;   /* GetRawScore (path rescoring) != x-drop score; tests only */
unsigned long orc_local_rescore_diffs(void) { return g_local_rescore_diffs; }

static LocalWork *lw_new(const orc_db *db)
{
  LocalWork *lw = (LocalWork *)calloc(1, sizeof *lw);
  lw->W = (unsigned)db->p.hsp_word_len;                       /* makedbsearcher.cpp:107-121: -hspw, 5 nt / 3 aa */
  lw->Alpha = db->alpha;
  lw->Dict = 1; for (unsigned i = 0; i < lw->W; ++i) lw->Dict *= lw->Alpha;
  lw->AlphaHi = lw->Dict / lw->Alpha;
  lw->Counts = (uint32_t *)calloc(lw->Dict, 4); lw->Counts2 = (uint32_t *)calloc(lw->Dict, 4);
  lw->Base = (uint32_t *)calloc(lw->Dict, 4);
  es_init(&lw->es, &db->p);
  orc_xdrop_params_init(&lw->xp, db->p.is_nucleo);
  lw->xp.match = db->p.match; lw->xp.mismatch = db->p.mismatch;
  lw->xp.local_open = db->p.local_open; lw->xp.local_ext = db->p.local_ext; lw->xp.xdrop = db->p.xdrop_g;
  return lw;
}
static void lw_free(LocalWork *lw)
{
  if (!lw) return;
  for (unsigned k = 0; k < lw->arcap; ++k) free(lw->ars[k].Path);
  free(lw->QueryWords); free(lw->QueryPosVec); free(lw->Counts); free(lw->Counts2); free(lw->Base);
  free(lw->TargetWords); free(lw->ars); free(lw->path); free(lw);
}

/* rolling words with wildcards as letter 0 (localaligner2.cpp:82-112, localmulti.cpp:31-56) */
static unsigned lw_words(const LocalWork *lw, const byte *c2l, const byte *S, unsigned L, uint32_t *Words)
{
  uint32_t Word = 0; const byte *Front = S, *Back = S; unsigned n = 0;
  for (unsigned i = 0; i + 1 < lw->W; ++i) { unsigned Letter = c2l[*Front++]; if (Letter >= lw->Alpha) Letter = 0; Word = Word * lw->Alpha + Letter; }
  for (unsigned Pos = lw->W - 1; Pos < L; ++Pos) {
    unsigned Letter = c2l[*Front++]; if (Letter >= lw->Alpha) Letter = 0;
    Word = Word * lw->Alpha + Letter;
    Words[n++] = Word;
    Letter = c2l[*Back++]; if (Letter >= lw->Alpha) Letter = 0;
    Word -= Letter * lw->AlphaHi;
  }
  return n;
}

/* LocalAligner2::SetQueryImpl localaligner2.cpp:62-145 (+ LocalAligner::SetQueryImpl localaligner.cpp:231-235) */
static void lw_set_query(LocalWork *lw, const orc_db *db, const byte *Q, unsigned QL)
{
  /* OnQueryDoneImpl of the previous query (localaligner2.cpp:147-158) */
  for (unsigned k = 0; k < lw->nwords; ++k) { lw->Counts[lw->QueryWords[k]] = 0; lw->Counts2[lw->QueryWords[k]] = 0; }
  lw->MinUngapped = (float)es_min_ungapped(&lw->es, QL);
  if (QL <= lw->W) { lw->nwords = 0; return; }               /* :72-73 (the reference leaves the old words; they are zeroed above) */
  if (QL > lw->qcap) { lw->qcap = QL + 256; lw->QueryWords = (uint32_t *)xrealloc(lw->QueryWords, lw->qcap * 4); lw->QueryPosVec = (uint32_t *)xrealloc(lw->QueryPosVec, lw->qcap * 4); }
  unsigned n = lw_words(lw, db->c2l, Q, QL, lw->QueryWords);
  lw->nwords = n;
  for (unsigned k = 0; k < n; ++k) ++lw->Counts2[lw->QueryWords[k]];
  unsigned Base = 0;
  for (unsigned k = 0; k < n; ++k) {
    uint32_t Word = lw->QueryWords[k]; unsigned c = lw->Counts2[Word];
    if (c == 0) continue;
    lw->Base[Word] = Base; lw->Counts2[Word] = 0; Base += c;
  }
  for (unsigned k = 0; k < n; ++k) {
    uint32_t Word = lw->QueryWords[k]; unsigned c = lw->Counts[Word];
    lw->Counts[Word] = c + 1;
    lw->QueryPosVec[lw->Base[Word] + c] = k;
  }
}

/* GetAnchor localaligner.cpp:11-58: best run of positive-scoring columns inside the ungapped segment */
static float lw_get_anchor(const float (*Sub)[256], const byte *Q, const byte *T, unsigned Loi, unsigned Loj, unsigned L,
                           unsigned *AncLoi, unsigned *AncLoj, unsigned *AncLen)
{
  unsigned Startk = UINT_MAX, BestStartk = UINT_MAX, Length = 0;
  float AnchorScore = 0.0f, BestScore = 0.0f;
  for (unsigned k = 0; k < L; ++k) {
    float Score = Sub[Q[Loi + k]][T[Loj + k]];
    if (Score > 0) {
      if (Startk == UINT_MAX) { Startk = k; AnchorScore = Score; } else AnchorScore += Score;
    } else {
      if (AnchorScore > BestScore) { BestScore = AnchorScore; BestStartk = Startk; Length = k - Startk; }
      Startk = UINT_MAX;
    }
  }
  if (AnchorScore > BestScore) { BestScore = AnchorScore; BestStartk = Startk; Length = L - Startk; }
  *AncLoi = Loi + BestStartk; *AncLoj = Loj + BestStartk; *AncLen = Length;
  return BestScore;
}

/* LocalAligner::AlignPos localaligner.cpp:101-222.  Returns 1 and fills *ar (path in lw->path) when an AR comes back. */
static int lw_align_pos(Work *w, LocalWork *lw, const byte *Q, unsigned QL, const byte *T, unsigned TL,
                        unsigned QueryPos, unsigned TargetPos, struct LAR *ar)
{
  const orc_db *db = w->db;
  const float (*Sub)[256] = (const float (*)[256])db->subst;
  const float XDropU = db->p.xdrop_u;
  float LeftScore = 0.0f, LeftTotal = 0.0f; unsigned LeftLength = 0, k = 0;
  int i = (int)QueryPos, j = (int)TargetPos;
  ++w->st.ungapped_calls;
  while (i >= 0 && j >= 0) {
    ++k;
    LeftTotal += Sub[Q[i]][T[j]];
    if (LeftTotal > LeftScore) { LeftScore = LeftTotal; LeftLength = k; }
    else if (LeftScore - LeftTotal > XDropU) break;
    --i; --j;
  }
  float RightScore = 0.0f, RightTotal = 0.0f; unsigned RightLength = 0;
  i = (int)QueryPos + 1; j = (int)TargetPos + 1; k = 0;
  while (i < (int)QL && j < (int)TL) {
    ++k;
    RightTotal += Sub[Q[i]][T[j]];
    if (RightTotal > RightScore) { RightScore = RightTotal; RightLength = k; }
    else if (RightScore - RightTotal > XDropU) break;
    ++i; ++j;
  }
  const float Score = LeftScore + RightScore;
  if (Score < lw->MinUngapped) return 0;
  unsigned Loi = (QueryPos + 1) - LeftLength, Loj = (TargetPos + 1) - LeftLength, SegLength = LeftLength + RightLength;
  unsigned AncLoi, AncLoj, AncLen;
  float AncRaw = lw_get_anchor(Sub, Q, T, Loi, Loj, SegLength, &AncLoi, &AncLoj, &AncLen);
  if (AncRaw <= 0.0f) return 0;
  if ((size_t)QL + TL + 8 > lw->pathcap) { lw->pathcap = (size_t)QL + TL + 256; lw->path = (char *)xrealloc(lw->path, lw->pathcap); }
  ugs_xdrop_job job; memset(&job, 0, sizeof job);
  job.anc_loi = AncLoi; job.anc_loj = AncLoj; job.anc_len = AncLen; job.mode = UGS_XDROP_ALIGN;
  ugs_xdrop_hsp hsp; uint64_t cells = 0;
  orc_xdrop_job(&lw->xp, (const char *)Q, QL, (const char *)T, TL, &job, &hsp, lw->path, &cells);
  w->st.dp_cells += cells; ++w->st.dp_calls;
  if (hsp.score <= 0.0f) return 0;
  if (es_evalue(&lw->es, (double)hsp.score, QL, 1) > (double)db->p.evalue) return 0;
  ar->Loi = hsp.loi; ar->Loj = hsp.loj; ar->Leni = hsp.leni; ar->Lenj = hsp.lenj; ar->Score = hsp.score;
  return 1;
}

/* HSPData::OverlapFract hsp.h:74-89 (unsigned products, Hi - Lo without the +1) */
static double lar_overlap(const struct LAR *a, const struct LAR *b)
{
  if (a->Leni == 0 || a->Lenj == 0) return 0.0;
  unsigned MaxLoi = a->Loi > b->Loi ? a->Loi : b->Loi, MaxLoj = a->Loj > b->Loj ? a->Loj : b->Loj;
  unsigned aHii = a->Loi + a->Leni - 1, bHii = b->Loi + b->Leni - 1, aHij = a->Loj + a->Lenj - 1, bHij = b->Loj + b->Lenj - 1;
  unsigned MinHii = aHii < bHii ? aHii : bHii, MinHij = aHij < bHij ? aHij : bHij;
  unsigned Ovi = (MinHii < MaxLoi) ? 0 : MinHii - MaxLoi, Ovj = (MinHij < MaxLoj) ? 0 : MinHij - MaxLoj;
  return (double)(Ovi * Ovj) / (double)(a->Leni * a->Lenj);
}

/* LocalAligner2::AlignMulti localmulti.cpp:9-118 */
static void lw_align_multi(Work *w, LocalWork *lw, const byte *Q, unsigned QL, const byte *T, unsigned TL)
{
  lw->nars = 0;
  if (TL < 2 * lw->W) return;
  if (TL > lw->tcap) { lw->tcap = TL + 256; lw->TargetWords = (uint32_t *)xrealloc(lw->TargetWords, lw->tcap * 4); }
  const unsigned TargetWordCount = lw_words(lw, w->db->c2l, T, TL, lw->TargetWords);
  for (unsigned TargetPos = 0; TargetPos < TargetWordCount;) {
    uint32_t TargetWord = lw->TargetWords[TargetPos];
    unsigned N = lw->Counts[TargetWord];
    int skipped = 0;
    for (unsigned i = 0; i < N; ++i) {
      unsigned QueryPos = lw->QueryPosVec[lw->Base[TargetWord] + i];
      struct LAR ar;
      if (!lw_align_pos(w, lw, Q, QL, T, TL, QueryPos, TargetPos, &ar)) continue;
      int keep = 1;                                           /* KeepAR localaligner2.cpp:228-246 */
      for (unsigned a = 0; a < lw->nars; ++a) if (lar_overlap(&ar, &lw->ars[a]) > 0.5) { keep = 0; break; }
      if (!keep) continue;
      if (lw->nars + 1 > lw->arcap) {
        unsigned nc = lw->arcap * 2 + 8;
        lw->ars = (struct LAR *)xrealloc(lw->ars, nc * sizeof *lw->ars);
        memset(lw->ars + lw->arcap, 0, (nc - lw->arcap) * sizeof *lw->ars);
        lw->arcap = nc;
      }
      struct LAR *dst = &lw->ars[lw->nars++];
      char *keepbuf = dst->Path;
      *dst = ar;
      size_t n = strlen(lw->path) + 1;
      dst->Path = (char *)xrealloc(keepbuf, n);
      memcpy(dst->Path, lw->path, n);
      unsigned NewTargetPos = ar.Loj + ar.Lenj - 1 + 1;       /* HSP.GetHij() + 1 */
      if (NewTargetPos > TargetPos) TargetPos = NewTargetPos; else ++TargetPos;
      skipped = 1;
      break;
    }
    if (!skipped) ++TargetPos;
  }
}

/* AlnParams::ScoreLocalPathIgnoreMask alnparams.cpp:447-500 (what AlignResult::GetRawScore reports) */
static float lw_rescore(const orc_db *db, const byte *A, const byte *B, const char *Path)
{
  const float (*Sub)[256] = (const float (*)[256])db->subst;
  float Score = 0.0f; char Last = 'M';
  for (const char *p = Path; *p; ++p) {
    if (*p == 'M') Score += Sub[toupper(*A++)][toupper(*B++)];
    else if (*p == 'D') { Score += (Last == 'M' ? db->p.local_open : db->p.local_ext); ++A; }
    else { Score += (Last == 'M' ? db->p.local_open : db->p.local_ext); ++B; }
    Last = *p;
  }
  return Score;
}

/* Accepter::IsAcceptLo accepter.cpp:24-91 for a local AR (coverages arscorer.cpp:122-154 local branches) */
static int is_accept_local(const ugs_params *p, const EStats *es, const ugs_hit *h)
{
  const unsigned m = p->filter_mask;
  if (p->id_set) {
    double FractId = h->aln_len == 0 ? 0.0 : (double)h->ids / (double)h->aln_len;
    if (FractId < p->id_accept) return 0;
    if ((m & UGS_F_MAXID) && FractId > (double)p->maxid) return 0;
  }
  if ((m & UGS_F_MINCOLS) && h->aln_len < p->mincols) return 0;
  if ((m & UGS_F_MAXGAPS) && h->gaps_int > p->maxgaps) return 0;
  if (es_evalue(es, (double)h->raw_score, h->ql, 1) > (double)p->evalue) return 0;
  if (m & (UGS_F_QUERY_COV | UGS_F_MAX_QUERY_COV)) {
    double Cov = (double)(h->qhi - h->qlo + 1) / (double)h->ql;
    if ((m & UGS_F_QUERY_COV) && Cov < (double)p->query_cov) return 0;
    if ((m & UGS_F_MAX_QUERY_COV) && Cov > (double)p->max_query_cov) return 0;
  }
  if (m & (UGS_F_TARGET_COV | UGS_F_MAX_TARGET_COV)) {
    double Cov = (double)(h->thi - h->tlo + 1) / (double)h->tl;
    if ((m & UGS_F_TARGET_COV) && Cov < (double)p->target_cov) return 0;
    if ((m & UGS_F_MAX_TARGET_COV) && Cov > (double)p->max_target_cov) return 0;
  }
  if ((m & UGS_F_MAXDIFFS) && h->mism + h->gaps_int > p->maxdiffs) return 0;
  if ((m & UGS_F_MINDIFFS) && h->mism + h->gaps_int < p->mindiffs) return 0;
  return 1;
}


static void search_strand_local(Work *w, LocalWork *lw, uint32_t qindex, const byte *q, unsigned QL, int strand, HitBuf *hb)
{
  orc_db *db = w->db;
  if (db->big) rank_big(w, q, QL); else rank_small(w, q, QL);
  lw_set_query(lw, db, q, QL);
  w->st.query_letters += QL;
  int AcceptCount = 0, RejectCount = 0;
  for (unsigned k = 0; k < w->ntop; ++k) {
    uint32_t t = w->cand_t[k];
    const byte *T = (const byte *)db->seqs + db->offs[t];
    unsigned TL = (unsigned)(db->offs[t + 1] - db->offs[t]);
    w->st.target_letters += TL; ++w->st.pairs_aligned;
    lw_align_multi(w, lw, q, QL, T, TL);
    int AnyAccepts = 0;                                       /* searcher.cpp:31-49 */
    for (unsigned a = 0; a < lw->nars; ++a) {
      const struct LAR *ar = &lw->ars[a];
      ugs_hit h; memset(&h, 0, sizeof h);
      fill_hit(db, ar->Path, q + ar->Loi, QL, T + ar->Loj, TL, &h);
      h.qlo = ar->Loi; h.qhi = ar->Loi + ar->Leni - 1; h.tlo = ar->Loj; h.thi = ar->Loj + ar->Lenj - 1;   /* m_HSP */
      h.raw_score = lw_rescore(db, q + ar->Loi, T + ar->Loj, ar->Path);
      if (h.raw_score != ar->Score) ++g_local_rescore_diffs;
      h.flags = UGS_HIT_LOCAL;
      if (is_accept_local(&db->p, &lw->es, &h)) {
        AnyAccepts = 1;
        h.query = qindex; h.target = t; h.strand = (uint32_t)strand;
        hb_push(hb, &h, ar->Path); ++w->st.hits;
      }
    }
    if (AnyAccepts) ++AcceptCount; else ++RejectCount;
    if (db->p.max_accepts > 0 && AcceptCount == db->p.max_accepts) break;
    if (db->p.max_rejects > 0 && RejectCount == db->p.max_rejects) break;
  }
}

/* Terminator::Terminate's -termid / -termidd tests (terminator.cpp:66-87) over the query's hits so far, both strands
 * (HitMgr::GetMinFractId / GetMaxFractId hitmgr.cpp:508-532: floats, start values 1 and 0) */
static int term_by_id(const ugs_params *p, const HitBuf *hb, unsigned qfirst)
{
  if (!(p->align_flags & (UGS_A_TERMID | UGS_A_TERMIDD)) || hb->nhits == qfirst) return 0;
  float MinId = 1.0f, MaxId = 0.0f;
  for (unsigned i = qfirst; i < hb->nhits; ++i) {
    const ugs_hit *h = &hb->hits[i];
    float f = (float)(h->aln_len == 0 ? 0.0 : (double)h->ids / (double)h->aln_len);
    if (f < MinId) MinId = f;
    if (f > MaxId) MaxId = f;
  }
  if ((p->align_flags & UGS_A_TERMID) && (double)MinId <= (double)p->termid) return 1;
  if ((p->align_flags & UGS_A_TERMIDD) && (double)(MaxId - MinId) > (double)p->termidd) return 1;
  return 0;
}

static void search_strand(Work *w, uint32_t qindex, const byte *q, unsigned QL, int strand, HitBuf *hb, unsigned qfirst)
{
  orc_db *db = w->db;
  if (db->big) rank_big(w, q, QL); else rank_small(w, q, QL);
  hf_set_a(w, q, QL);
  w->st.query_letters += QL;
  int AcceptCount = 0, RejectCount = 0;
  for (unsigned k = 0; k < w->ntop; ++k) {
    uint32_t t = w->cand_t[k];
    const byte *T = (const byte *)db->seqs + db->offs[t];
    unsigned TL = (unsigned)(db->offs[t + 1] - db->offs[t]);
    w->st.target_letters += TL; ++w->st.pairs_aligned;
    if (db->p.pair_mask && reject_pair(db, qindex, q, QL, t, T, TL)) {
      /* Big path: SetTarget fails -> Terminate(HM, false) (udbusortedsearcherbig.cpp:118-127); small path: SetTarget's
       * result is ignored and AlignPos returns without touching the terminator (udbusortedsearcher.cpp:145-147, searcher.cpp:63-67) */
      --w->st.pairs_aligned; w->st.target_letters -= TL;
      if (!db->big) continue;
      if (term_by_id(&db->p, hb, qfirst)) break;
      ++RejectCount;
      if (db->p.max_rejects > 0 && RejectCount == db->p.max_rejects) break;
      continue;
    }
    float fid;
    int aligned = global_align(w, T, TL, &fid);
    int Accept = 0;
    if (aligned) {
      ugs_hit h; memset(&h, 0, sizeof(h));
      fill_hit(db, w->path, q, QL, T, TL, &h);
      Accept = is_accept_lo(&db->p, &h);
      if (Accept && (db->p.filter_mask & UGS_F_ABSKEW)) {       /* accepter.cpp:89-90, GetAbSkew arscorer.cpp:809-816 */
        unsigned QS = db->q_size ? db->q_size[qindex] : UINT_MAX, TS = db->t_size ? db->t_size[t] : UINT_MAX;
        if ((double)TS / (double)QS < (double)db->p.abskew) Accept = 0;
      }
      if (Accept) { h.query = qindex; h.target = t; h.strand = (uint32_t)strand; hb_push(hb, &h, w->path); ++w->st.hits; }
    }
    if (term_by_id(&db->p, hb, qfirst)) break;
    if (Accept) ++AcceptCount; else ++RejectCount;
    if (db->p.max_accepts > 0 && AcceptCount == db->p.max_accepts) break;
    if (db->p.max_rejects > 0 && RejectCount == db->p.max_rejects) break;
  }
}

typedef struct {
  orc_db *db; const char *qseqs; const uint64_t *qoffs; uint32_t q0, q1;
  HitBuf hb; uint32_t *nhits; orc_stats st;
} Job;

static void *job_run(void *arg)
{
  Job *J = (Job *)arg;
  orc_db *db = J->db;
  Work *w = work_new(db);
  LocalWork *lw = db->p.local ? lw_new(db) : NULL;
  char *rc = NULL; size_t rccap = 0;
  for (uint32_t qi = J->q0; qi < J->q1; ++qi) {
    const byte *q = (const byte *)J->qseqs + J->qoffs[qi];
    unsigned QL = (unsigned)(J->qoffs[qi + 1] - J->qoffs[qi]);
    unsigned first = J->hb.nhits;
    if (lw) search_strand_local(w, lw, qi, q, QL, 0, &J->hb); else search_strand(w, qi, q, QL, 0, &J->hb, first);
    if (db->p.strand_both && db->p.is_nucleo) {
      if (rccap < QL + 1) { rccap = QL + 256; rc = (char *)xrealloc(rc, rccap); }
      orc_revcomp((const char *)q, QL, rc);
      if (lw) search_strand_local(w, lw, qi, (const byte *)rc, QL, 1, &J->hb); else search_strand(w, qi, (const byte *)rc, QL, 1, &J->hb, first);
    }
    unsigned n = J->hb.nhits - first;
    J->nhits[qi] = n;
    for (unsigned i = 0; i < n; ++i) J->hb.hits[first + i].flags |= i << UGS_HIT_ORDER_SHIFT;   /* append order, see include/ugs.h */
    if (n > 1) {     /* hitmgr.cpp:477-483 Sort by AlignResult::GetScore desc: float(FractId), local: float(raw score) */
      float *sc = (float *)malloc(n * sizeof(float));
      unsigned *ord = (unsigned *)malloc(n * sizeof(unsigned));
      ugs_hit *tmp = (ugs_hit *)malloc(n * sizeof(ugs_hit));
      for (unsigned i = 0; i < n; ++i) {
        const ugs_hit *h = &J->hb.hits[first + i];
        sc[i] = lw ? h->raw_score : (float)(h->aln_len == 0 ? 0.0 : (double)h->ids / (double)h->aln_len);
        ord[i] = i; tmp[i] = *h;
      }
      qs_order_desc(sc, 0, (int)n - 1, ord);
      for (unsigned i = 0; i < n; ++i) J->hb.hits[first + i] = tmp[ord[i]];
      free(sc); free(ord); free(tmp);
    }
  }
  J->st = w->st;
  free(rc);
  lw_free(lw);
  work_free(w);
  return NULL;
}

int orc_search_batch(orc_db *db, const char *qseqs, const uint64_t *qoffs, uint32_t nq,
                     ugs_hit *hits, uint64_t hits_cap, uint32_t *nhits_per_query,
                     uint32_t *cigar_pool, uint64_t cigar_cap, uint64_t *cigar_used, int nthreads)
{
  if (nthreads < 1) nthreads = 1;
  if ((uint32_t)nthreads > nq && nq > 0) nthreads = (int)nq;
  Job *jobs = (Job *)calloc((size_t)nthreads, sizeof(Job));
  pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
  for (int k = 0; k < nthreads; ++k) {
    jobs[k].db = db; jobs[k].qseqs = qseqs; jobs[k].qoffs = qoffs; jobs[k].nhits = nhits_per_query;
    jobs[k].q0 = (uint32_t)((uint64_t)nq * k / nthreads);
    jobs[k].q1 = (uint32_t)((uint64_t)nq * (k + 1) / nthreads);
  }
  if (nthreads == 1) job_run(&jobs[0]);
  else {
    for (int k = 0; k < nthreads; ++k) pthread_create(&th[k], NULL, job_run, &jobs[k]);
    for (int k = 0; k < nthreads; ++k) pthread_join(th[k], NULL);
  }
  int rc = UGS_OK;
  uint64_t nh = 0, nc = 0;
  memset(&db->stats, 0, sizeof(db->stats));
  for (int k = 0; k < nthreads; ++k) {
    HitBuf *hb = &jobs[k].hb;
    if (nh + hb->nhits > hits_cap || nc + hb->ncig > cigar_cap) rc = UGS_E_CAPACITY;
    if (rc == UGS_OK) {
      for (unsigned i = 0; i < hb->nhits; ++i) { hits[nh + i] = hb->hits[i]; hits[nh + i].cigar_off += nc; }
      if (hb->ncig) memcpy(cigar_pool + nc, hb->cig, hb->ncig * 4);
    }
    nh += hb->nhits; nc += hb->ncig;
    orc_stats *a = &db->stats, *b = &jobs[k].st;
    a->postings += b->postings; a->query_letters += b->query_letters; a->target_letters += b->target_letters;
    a->pairs_aligned += b->pairs_aligned; a->dp_cells += b->dp_cells; a->hits += b->hits;
    a->ungapped_calls += b->ungapped_calls; a->dp_calls += b->dp_calls;
    free(hb->hits); free(hb->cig);
  }
  if (cigar_used) *cigar_used = nc;
  free(jobs); free(th);
  return rc;
}


/* ================================================================== cluster_fast (BASELINE config C3)
 * clusterfast.cpp:81-133 ClusterFast: DerepFull (derepfull.cpp:130-212, at -threads 1: uniques in input order of their first
 * member, derepresult.cpp:403-480) -> serial loop over the uniques in input order (-sort unset) -> Searcher::Search against
 * the centroids found so far (terminator 1 accept / 8 rejects, terminator.cpp:10-14; small -> Big latch
 * udbusortedsearcher.cpp:39-58) -> ClusterSink::OnQueryDone (clustersink.cpp:306-359): GetTopHit => member of that hit's
 * cluster, no hit => new centroid, appended to the SeqDB and to every row of its distinct words
 * (UDBData::AddSIToDB_CopyData udbbuild.cpp:286-291, AddSeqNoncoded :256-284, AddWord/GrowRow :74-128).
 * No masking anywhere on this path: SeqDB::FromFastx keeps the letters as read (lower case voids a word). */

/* SeqEq / SeqEqRC (seqhash.cpp:44-69): case-insensitive */
static int cl_seq_eq(const byte *a, const byte *b, unsigned L)
{
  for (unsigned i = 0; i < L; ++i) if (toupper(a[i]) != toupper(b[i])) return 0;
  return 1;
}
static int cl_seq_eq_rc(const byte *a, const byte *b, unsigned L)
{
  for (unsigned i = 0; i < L; ++i) if (toupper(a[i]) != toupper(g_comp[b[L - i - 1]])) return 0;
  return 1;
}
/* SeqHash32 / SeqHashRC32 (seqhash.cpp:6-34); only the grouping matters at -threads 1, the function is kept all the same */
static uint32_t cl_hash(const byte *s, unsigned L, int rc)
{
  unsigned a = 63689, b = 378551; uint32_t h = 0;
  for (unsigned k = 0; k < L; ++k) { byte c = rc ? g_comp[s[L - k - 1]] : s[k]; h = h * a + (uint32_t)toupper(c); a *= b; }
  return h;
}

/* derepfull.cpp:23-128 Thread() with one thread + derepresult.cpp:403-480: seq_unique[i] = index of input i's unique
 * (uniques numbered by first appearance), uniq_seed[u] = input index of unique u's first member.  Returns the count. */
uint32_t orc_derep_full(const char *seqs, const uint64_t *offs, uint32_t nseq, int revcomp, uint32_t *seq_unique, uint32_t *uniq_seed)
{
  init_tables();
  uint64_t slots = (uint64_t)nseq * 8 + 7;
  uint32_t *tab = (uint32_t *)malloc(slots * 4);
  for (uint64_t i = 0; i < slots; ++i) tab[i] = UINT32_MAX;
  uint32_t nu = 0;
  for (uint32_t i = 0; i < nseq; ++i) {
    const byte *q = (const byte *)seqs + offs[i];
    unsigned L = (unsigned)(offs[i + 1] - offs[i]);
    uint32_t h = cl_hash(q, L, 0);
    if (revcomp) { uint32_t h2 = cl_hash(q, L, 1); if (h2 < h) h = h2; }
    uint64_t k = h % slots;
    for (;;) {
      uint32_t u = tab[k];
      if (u == UINT32_MAX) { tab[k] = nu; uniq_seed[nu] = i; seq_unique[i] = nu; ++nu; break; }
      uint32_t si = uniq_seed[u];
      unsigned UL = (unsigned)(offs[si + 1] - offs[si]);
      if (UL == L) {
        const byte *us = (const byte *)seqs + offs[si];
        if (cl_seq_eq(q, us, L) || (revcomp && cl_seq_eq_rc(q, us, L))) { seq_unique[i] = u; break; }
      }
      k = (k + 1) % slots;
    }
  }
  free(tab);
  return nu;
}

/* UDBData::AddSIToDB_CopyData (udbbuild.cpp:286-291): letters appended as they are, every distinct valid word's row gets
 * the new index at its end (AddSeqNoncoded :256-284: SetTargetWords + SetTargetUniqueWords) */
static void cl_append(orc_db *db, const byte *seq, unsigned L, byte *stampw)
{
  uint64_t tot = db->offs[db->nseq];
  if (tot + L > db->seq_cap) { db->seq_cap = (tot + L) * 2 + 1024; db->seqs = (char *)xrealloc(db->seqs, db->seq_cap); }
  if (db->nseq + 2 > db->off_cap) { db->off_cap = db->off_cap * 2 + 1024; db->offs = (uint64_t *)xrealloc(db->offs, (size_t)db->off_cap * 8); }
  memcpy(db->seqs + tot, seq, L);
  db->offs[db->nseq + 1] = tot + L;
  const uint32_t t = db->nseq;
  const int W = db->p.word_len;
  if (L >= (unsigned)W) {
    for (unsigned pos = 0; pos + W <= L; ++pos) {
      uint32_t w = seq_to_word(db, seq + pos);
      if (w == BAD_WORD || stampw[w]) continue;
      stampw[w] = 1;
      if (db->dyn_size[w] == db->dyn_cap[w]) {
        db->dyn_cap[w] = db->dyn_cap[w] ? db->dyn_cap[w] * 2 : 16;
        db->dyn_rows[w] = (uint32_t *)xrealloc(db->dyn_rows[w], (size_t)db->dyn_cap[w] * 4);
      }
      db->dyn_rows[w][db->dyn_size[w]++] = t;
    }
    for (unsigned pos = 0; pos + W <= L; ++pos) { uint32_t w = seq_to_word(db, seq + pos); if (w != BAD_WORD) stampw[w] = 0; }
  }
  if (L > db->maxlen) db->maxlen = L;
  db->nseq = t + 1;
}

/* sort.h:63-103 QuickSortOrderRecurse<unsigned, Desc=true> (ClusterSink::GetClusterSizeOrder clustersink.cpp:449-458) */
static void qs_order_desc_u(const uint32_t *V, int left, int right, uint32_t *Order)
{
  int i = left, j = right;
  uint32_t pivot = V[Order[(left + right) / 2]];
  while (i <= j) {
    while (V[Order[i]] > pivot) i++;
    while (V[Order[j]] < pivot) j--;
    if (i <= j) { uint32_t t = Order[i]; Order[i] = Order[j]; Order[j] = t; i++; j--; }
  }
  if (left < j) qs_order_desc_u(V, left, j, Order);
  if (i < right) qs_order_desc_u(V, i, right, Order);
}
void orc_order_desc_u32(const uint32_t *values, uint32_t n, uint32_t *order)
{
  for (uint32_t i = 0; i < n; ++i) order[i] = i;
  if (n) qs_order_desc_u(values, 0, (int)n - 1, order);
}

/* The greedy loop.  In: the input sequences (letters as read).  Out, all caller-allocated with nseq entries unless noted:
 *   seq_unique[i]   unique (derep cluster) of input i;   uniq_seed[u] input index of unique u's first member
 *   uniq_cluster[u] cluster of unique u;  uniq_nhits[u] hits of unique u (0 = it founded its cluster, else 1, or 2 with
 *   -strand both);  centroid_uniq[c] the unique that founded cluster c;  cluster_size[c] members incl. duplicates
 *   hits[]          all hits grouped by unique in order, each group in HitMgr::Sort order; .query = unique, .target = cluster
 * p: terminator 1/8 and dbmask 2 are the caller's job (orc_params_set_cluster). */
int orc_cluster_fast(const ugs_params *p, const char *seqs, const uint64_t *offs, uint32_t nseq,
                     uint32_t *seq_unique, uint32_t *uniq_seed, uint32_t *n_unique,
                     uint32_t *uniq_cluster, uint32_t *uniq_nhits, uint32_t *centroid_uniq, uint32_t *cluster_size,
                     uint32_t *n_clusters, ugs_hit *hits, uint64_t hits_cap, uint32_t *cigar_pool, uint64_t cigar_cap,
                     uint64_t *n_hits, uint64_t *cigar_used)
{
  return orc_cluster_fast_sorted(p, seqs, offs, nseq, 0, NULL, 0, seq_unique, uniq_seed, n_unique, uniq_cluster, uniq_nhits, centroid_uniq,
                                 cluster_size, n_clusters, hits, hits_cap, cigar_pool, cigar_cap, n_hits, cigar_used);
}

/* -sort length | size (GetSeqOrder clusterfast.cpp:37-79: QuickSortOrderDesc over the uniques' seed lengths / DerepResult::GetSumSizeIn
 * derepresult.cpp:211-225, which reads ;size= with default 1 whether or not -sizein is set) and -sizein (ClusterSink::GetSize
 * clustersink.cpp:119-150: cluster sizes sum the annotations, a label without one is fatal).  size_in[nseq]: the ;size= value
 * of every input label (GetSizeFromLabel label.cpp:152-161), UINT32_MAX = none.  The uniques are renumbered in processing
 * order (= ClusterFast's loop order clusterfast.cpp:113-123), so uniq_seed / uniq_* / hits.query all follow it. */
int orc_cluster_fast_sorted(const ugs_params *p, const char *seqs, const uint64_t *offs, uint32_t nseq,
                     int sort_mode, const uint32_t *size_in, int sizein,
                     uint32_t *seq_unique, uint32_t *uniq_seed, uint32_t *n_unique,
                     uint32_t *uniq_cluster, uint32_t *uniq_nhits, uint32_t *centroid_uniq, uint32_t *cluster_size,
                     uint32_t *n_clusters, ugs_hit *hits, uint64_t hits_cap, uint32_t *cigar_pool, uint64_t cigar_cap,
                     uint64_t *n_hits, uint64_t *cigar_used)
{
  uint64_t zero = 0;
  orc_db *db = NULL;
  int rc = orc_db_create(p, "", &zero, 0, &db);
  if (rc != UGS_OK) return rc;
  db->dyn_rows = (uint32_t **)calloc(db->slots, sizeof(uint32_t *));
  db->dyn_size = (uint32_t *)calloc(db->slots, 4);
  db->dyn_cap = (uint32_t *)calloc(db->slots, 4);
  db->off_cap = 1;
  db->big = 0;
  const int revcomp = p->strand_both && p->is_nucleo;
  const uint32_t nu = orc_derep_full(seqs, offs, nseq, revcomp, seq_unique, uniq_seed);
  *n_unique = nu;
  if (sizein) {
    if (!size_in) { orc_db_destroy(db); return UGS_E_ARG; }
    for (uint32_t i = 0; i < nseq; ++i) if (size_in[i] == UINT32_MAX) { orc_db_destroy(db); return UGS_E_ARG; }   /* "Missing size= in >..." */
  }
  if (sort_mode == 1 || sort_mode == 2) {
    uint32_t *v = (uint32_t *)calloc(nu ? nu : 1, 4), *ord = (uint32_t *)calloc(nu ? nu : 1, 4), *rank = (uint32_t *)calloc(nu ? nu : 1, 4);
    uint32_t *seed2 = (uint32_t *)calloc(nu ? nu : 1, 4);
    if (sort_mode == 1) for (uint32_t u = 0; u < nu; ++u) v[u] = (uint32_t)(offs[uniq_seed[u] + 1] - offs[uniq_seed[u]]);
    else for (uint32_t i = 0; i < nseq; ++i) v[seq_unique[i]] += (size_in && size_in[i] != UINT32_MAX) ? size_in[i] : 1;
    orc_order_desc_u32(v, nu, ord);
    for (uint32_t k = 0; k < nu; ++k) { rank[ord[k]] = k; seed2[k] = uniq_seed[ord[k]]; }
    for (uint32_t i = 0; i < nseq; ++i) seq_unique[i] = rank[seq_unique[i]];
    memcpy(uniq_seed, seed2, (size_t)nu * 4);
    free(v); free(ord); free(rank); free(seed2);
  } else if (sort_mode != 0) { orc_db_destroy(db); return UGS_E_ARG; }
  uint32_t *usize = (uint32_t *)calloc(nu ? nu : 1, 4);          /* ClusterSink::GetSize: member count, or the summed annotations */
  for (uint32_t i = 0; i < nseq; ++i) usize[seq_unique[i]] += sizein ? size_in[i] : 1;
  Work *w = work_new_cap(db, nu);
  byte *stampw = (byte *)calloc(db->slots, 1);
  HitBuf hb; memset(&hb, 0, sizeof(hb));
  char *rcbuf = NULL; size_t rccap = 0;
  uint32_t nc = 0;
  for (uint32_t u = 0; u < nu; ++u) {
    const uint32_t si = uniq_seed[u];
    const byte *q = (const byte *)seqs + offs[si];
    const unsigned QL = (unsigned)(offs[si + 1] - offs[si]);
    const unsigned first = hb.nhits;
    /* UDBUsortedSearcher::SetQueryImpl (udbusortedsearcher.cpp:39-58): the Big latch, tested once per strand search */
    if (!db->big && db->nseq > p->big) { db->big = 1; memset(w->U, 0, (size_t)(nu ? nu : 1) * 4); }   /* :45-48 zeroes U at the latch */
    search_strand(w, u, q, QL, 0, &hb, first);
    if (revcomp) {
      if (rccap < QL + 1) { rccap = QL + 256; rcbuf = (char *)xrealloc(rcbuf, rccap); }
      orc_revcomp((const char *)q, QL, rcbuf);
      if (!db->big && db->nseq > p->big) { db->big = 1; memset(w->U, 0, (size_t)(nu ? nu : 1) * 4); }
      search_strand(w, u, (const byte *)rcbuf, QL, 1, &hb, first);
    }
    const unsigned n = hb.nhits - first;
    uniq_nhits[u] = n;
    if (n == 0) {                                               /* clustersink.cpp:318-329 */
      cl_append(db, q, QL, stampw);
      centroid_uniq[nc] = u; cluster_size[nc] = usize[u]; uniq_cluster[u] = nc; ++nc;
    } else {
      /* HitMgr::GetTopHit hitmgr.cpp:400-420: best float(FractId), ties to the smaller target index, else the earlier hit */
      unsigned top = first; float tops = 0; uint32_t mint = 0;
      for (unsigned i = first; i < hb.nhits; ++i) {
        const ugs_hit *h = &hb.hits[i];
        float sc = (float)(h->aln_len == 0 ? 0.0 : (double)h->ids / (double)h->aln_len);
        if (i == first || sc > tops || (sc == tops && h->target < mint)) { top = i; tops = sc; mint = h->target; }
      }
      const uint32_t c = hb.hits[top].target;
      uniq_cluster[u] = c; cluster_size[c] += usize[u];
      for (unsigned i = 0; i < n; ++i) hb.hits[first + i].flags |= i << UGS_HIT_ORDER_SHIFT;
      if (n > 1) {                                              /* HitMgr::Sort hitmgr.cpp:477-483 */
        float sc[2]; unsigned ord[2] = {0, 1}; ugs_hit tmp[2];
        if (n != 2) abort();
        for (unsigned i = 0; i < 2; ++i) { const ugs_hit *h = &hb.hits[first + i]; sc[i] = (float)(h->aln_len == 0 ? 0.0 : (double)h->ids / (double)h->aln_len); tmp[i] = *h; }
        qs_order_desc(sc, 0, 1, ord);
        for (unsigned i = 0; i < 2; ++i) hb.hits[first + i] = tmp[ord[i]];
      }
    }
  }
  *n_clusters = nc;
  if (n_hits) *n_hits = hb.nhits;
  if (cigar_used) *cigar_used = hb.ncig;
  if (hb.nhits > hits_cap || hb.ncig > cigar_cap) rc = UGS_E_CAPACITY;
  else {
    if (hb.nhits) memcpy(hits, hb.hits, (size_t)hb.nhits * sizeof(ugs_hit));
    if (hb.ncig) memcpy(cigar_pool, hb.cig, (size_t)hb.ncig * 4);
  }
  free(hb.hits); free(hb.cig); free(rcbuf); free(stampw); free(usize);
  work_free(w);
  orc_db_destroy(db);
  return rc;
}

/* ------------------------------------------------------------------ stage entry points */

int orc_rank(orc_db *db, const char *q, uint32_t ql, uint32_t *cand, uint32_t *cnt, uint32_t cap)
{
  Work *w = work_new(db);
  if (db->big) rank_big(w, (const byte *)q, ql); else rank_small(w, (const byte *)q, ql);
  unsigned n = w->ntop;
  for (unsigned k = 0; k < n && k < cap; ++k) { cand[k] = w->cand_t[k]; cnt[k] = w->cand_c[k]; }
  work_free(w);
  return (int)n;
}

int orc_align_pair(orc_db *db, const char *q, uint32_t ql, const char *t, uint32_t tl,
                   char *path, uint32_t cap, float *hsp_fract_id)
{
  Work *w = work_new(db);
  hf_set_a(w, (const byte *)q, ql);
  int ok = global_align(w, (const byte *)t, tl, hsp_fract_id);
  if (ok) {
    size_t n = strlen(w->path);
    if (n + 1 > cap) ok = -1; else memcpy(path, w->path, n + 1);
  } else if (cap) path[0] = 0;
  work_free(w);
  return ok;
}

float orc_viterbi_band(orc_db *db, const char *a, uint32_t la, const char *b, uint32_t lb,
                       uint32_t band, const float *pen, char *path, uint32_t cap, uint64_t *cells)
{
  Work *w = work_new(db);
  AlnPen P;
  P.OpenA = pen[0]; P.OpenB = pen[1]; P.ExtA = pen[2]; P.ExtB = pen[3];
  P.LOpenA = pen[4]; P.LOpenB = pen[5]; P.LExtA = pen[6]; P.LExtB = pen[7];
  P.ROpenA = pen[8]; P.ROpenB = pen[9]; P.RExtA = pen[10]; P.RExtB = pen[11];
  path_alloc(w, (size_t)la + lb + 2);
  float s = viterbi_main_diag(w, (const byte *)a, la, (const byte *)b, lb, band, &P, w->path);
  size_t n = strlen(w->path);
  if (n + 1 <= cap) memcpy(path, w->path, n + 1); else if (cap) path[0] = 0;
  if (cells) *cells = w->st.dp_cells;
  work_free(w);
  return s;
}

/* ------------------------------------------------------------------ writers */

/* blast6out.cpp:27-80: qstart..send are 1..QL / 1..TL (global m_HSP = whole sequences,
 * alignresult.cpp:138-145); with a rev-comp query sstart/send swap (arscorer.cpp:754-757) */
int orc_format_blast6(const ugs_hit *h, const char *qlabel, const char *tlabel, char *buf, int cap)
{
  double FractId = h->aln_len == 0 ? 0.0 : (double)h->ids / (double)h->aln_len;
  double PctId = 100.0 * FractId;
  unsigned TLo = 1, THi = h->tl;
  if (h->strand) { TLo = h->tl; THi = 1; }
  return snprintf(buf, (size_t)cap, "%s\t%s\t%.1f\t%u\t%u\t%u\t%u\t%u\t%u\t%u\t*\t*\n", qlabel, tlabel,
                  PctId, h->aln_len, h->mism, h->opens, 1u, h->ql, TLo, THi);
}

/* blast6out.cpp:27-80 for a local hit: 1-based HSP coordinates, target pair swapped for a reverse-complemented
 * query (arscorer.cpp:688-806), e-value %.2g and bit score %.1f */
int orc_format_blast6_local(const ugs_params *p, const ugs_hit *h, const char *qlabel, const char *tlabel, char *buf, int cap)
{
  double FractId = h->aln_len == 0 ? 0.0 : (double)h->ids / (double)h->aln_len;
  double PctId = 100.0 * FractId;
  unsigned QLo = h->qlo + 1, QHi = h->qhi + 1, TLo = h->tlo + 1, THi = h->thi + 1;
  if (h->strand) { QLo = h->ql - h->qhi; QHi = h->ql - h->qlo; unsigned t = TLo; TLo = THi; THi = t; }
  double E, Bits;
  orc_local_evalue(p, (double)h->raw_score, h->ql, &E, &Bits);
  return snprintf(buf, (size_t)cap, "%s\t%s\t%.1f\t%u\t%u\t%u\t%u\t%u\t%u\t%u\t%.2g\t%.1f\n", qlabel, tlabel,
                  PctId, h->aln_len, h->mism, h->opens, QLo, QHi, TLo, THi, E, Bits);
}

/* outputuc.cpp:45-93 + comppath.cpp:7-48 */
int orc_format_uc_hit(const ugs_hit *h, const uint32_t *cigar_pool, int is_nucleo,
                      const char *qlabel, const char *tlabel, char *buf, int cap)
{
  double FractId = h->aln_len == 0 ? 0.0 : (double)h->ids / (double)h->aln_len;
  double PctId = 100.0 * FractId;
  char strand = !is_nucleo ? '.' : (h->strand ? '-' : '+');
  int n = snprintf(buf, (size_t)cap, "H\t%u\t%u\t%.1f\t%c\t%u\t%u\t", h->target, h->ql, PctId, strand, 0u, 0u);
  for (uint32_t k = 0; k < h->cigar_len; ++k) {
    uint32_t r = cigar_pool[h->cigar_off + k];
    char op = "MDI"[r & 3];
    uint32_t len = r >> 2;
    if (len == 1) n += snprintf(n < cap ? buf + n : NULL, n < cap ? (size_t)(cap - n) : 0, "%c", op);
    else n += snprintf(n < cap ? buf + n : NULL, n < cap ? (size_t)(cap - n) : 0, "%u%c", len, op);
  }
  n += snprintf(n < cap ? buf + n : NULL, n < cap ? (size_t)(cap - n) : 0, "\t%s\t%s\n", qlabel, tlabel);
  return n;
}

/* outputuc.cpp:10-23 */
int orc_format_uc_nohit(uint32_t ql, const char *qlabel, char *buf, int cap)
{
  return snprintf(buf, (size_t)cap, "N\t*\t%u\t*\t.\t*\t*\t*\t%s\t*\n", ql, qlabel);
}

/* ================================================================== gapped x-drop (8a X1-X3)
 * Restatement of the local aligner's extension kernel.  Same float arithmetic, same
 * tie rules, same window bookkeeping as the reference; arrays are explicit instead of XDPMem. */

#define XD_MAXL 4096u                 /* xdpmem.h:6 g_MaxL */
#define XD_NEG  (-9e9f)               /* mx.h:12 MINUS_INFINITY */
#define XD_DM 0x01                    /* tracebit.h:4-7 */
#define XD_IM 0x02
#define XD_MD 0x04
#define XD_MI 0x08

typedef struct XdMem {
  float *mrow_base, *drow;            /* mrow = mrow_base + 1 so mrow[-1] exists (xdropfwdmem.cpp:393) */
  byte *tb;                           /* (LA+1) x (LB+1) */
  size_t tb_cap; unsigned row_cap;
  byte *reva, *revb; unsigned rev_cap;
  uint64_t cells;
} XdMem;

static void xd_alloc(XdMem *m, unsigned LA, unsigned LB)
{
  size_t need = (size_t)(LA + 1) * (LB + 1);
  if (need > m->tb_cap) { m->tb = (byte *)xrealloc(m->tb, need); m->tb_cap = need; }
  if (LB + 4 > m->row_cap) {
    m->row_cap = LB + 4;
    m->mrow_base = (float *)xrealloc(m->mrow_base, sizeof(float) * (m->row_cap + 1));
    m->drow = (float *)xrealloc(m->drow, sizeof(float) * (m->row_cap + 1));
  }
  unsigned L = LA > LB ? LA : LB;
  if (L > m->rev_cap) {
    m->rev_cap = L;
    m->reva = (byte *)xrealloc(m->reva, L);
    m->revb = (byte *)xrealloc(m->revb, L);
  }
}

static void xd_free(XdMem *m)
{
  free(m->mrow_base); free(m->drow); free(m->tb); free(m->reva); free(m->revb);
  memset(m, 0, sizeof *m);
}

static unsigned long g_xd_wipes = 0;       /* how often the insert-branch re-initialisation hit a live cell */
unsigned long orc_xdrop_wipes(void) { return g_xd_wipes; }

static unsigned xd_min(unsigned a, unsigned b) { return a < b ? a : b; }
static unsigned xd_max(unsigned a, unsigned b) { return a > b ? a : b; }

/* XDropFwdFastMem xdropfwdmem.cpp:344-749 (+ traceback :271-342).  path receives the
 * forward M/D/I string, NUL-terminated (needs LA+LB+2 bytes). */
static float xd_fwd(XdMem *mem, const float (*Sub)[256], float Open, float Ext,
                    const byte *A, unsigned LA, const byte *B, unsigned LB, float X,
                    unsigned *Leni, unsigned *Lenj, char *path)
{
  if (LA == 1 || LB == 1) {                               /* :360-368 */
    *Leni = 1; *Lenj = 1; path[0] = 'M'; path[1] = 0;
    return Sub[A[0]][B[0]];
  }
  xd_alloc(mem, LA, LB);
  const float AbsOpen = -Open, AbsExt = -Ext;
  const size_t W = (size_t)LB + 1;
  byte *TB = mem->tb;
  float *Mrow = mem->mrow_base + 1, *Drow = mem->drow;
  Mrow[-1] = XD_NEG;
  Drow[0] = XD_NEG; Drow[1] = XD_NEG;

  float Best = Sub[A[0]][B[0]];                            /* :404 */
  unsigned Besti = 0, Bestj = 0, prev_jlo = 0, prev_jhi = 0, jlo = 1, jhi = 1;
  float M0 = Best;
  for (unsigned i = 1; i < LA; ++i) {
    if (jlo == prev_jlo) { Mrow[jlo - 1] = XD_NEG; Drow[jlo] = XD_NEG; }            /* :421-430 */
    unsigned endj = xd_min(prev_jhi + 1, LB);
    for (unsigned j = endj + 1; j <= xd_min(jhi + 1, LB); ++j) { Mrow[j - 1] = XD_NEG; Drow[j] = XD_NEG; }
    unsigned next_jlo = UINT32_MAX, next_jhi = UINT32_MAX;
    const float *MxRow = Sub[A[i]];
    float I0 = XD_NEG;
    byte *TBrow = TB + (size_t)i * W;
    for (unsigned j = jlo; j <= jhi; ++j) {
      ++mem->cells;
      byte bits = 0;
      const float SavedM0 = M0;
      /* match :478-560 */
      float xM = M0;
      if (Drow[j] > xM) { xM = Drow[j]; bits = XD_DM; }
      if (I0 > xM) { xM = I0; bits = XD_IM; }
      M0 = Mrow[j];
      float s = xM + MxRow[B[j]];
      Mrow[j] = s;
      float h = s - Best + X;
      if (h > 0) { next_jlo = xd_min(next_jlo, j + 1); next_jhi = j + 1; }
      if (h > AbsOpen) next_jlo = xd_min(next_jlo, j);
      if (h > AbsExt && j == jhi && jhi + 1 < LB) {
        ++jhi;
        unsigned new_endj = xd_max(xd_min(jhi + 1, LB), endj);
        for (unsigned j2 = endj + 1; j2 <= new_endj; ++j2) {
          if (j2 - 1 > j) Mrow[j2 - 1] = XD_NEG;            /* :538-546: Mrow[j] already holds row i+1 */
          Drow[j2] = XD_NEG;
        }
        endj = new_endj;
      }
      if (s >= Best) { Best = s; Besti = i; Bestj = j; }  /* :555 ties prefer later cells */
      /* delete :563-589 */
      if (j != jlo) {
        float md = SavedM0 + Open;
        Drow[j] += Ext;
        if (md >= Drow[j]) { Drow[j] = md; bits |= XD_MD; }
        float hd = Drow[j] - Best + X;
        if (hd > 0) { next_jlo = xd_min(next_jlo, j - 1); next_jhi = xd_max(next_jhi, j - 1); }
      }
      /* insert :592-640 */
      {
        float mi = SavedM0 + Open;
        I0 += Ext;
        if (mi >= I0) { I0 = mi; bits |= XD_MI; }
        float hi = I0 - Best + X;
        if (hi > 0) { next_jlo = xd_min(next_jlo, j + 1); next_jhi = j + 1; }
        if (hi > AbsExt && j == jhi && jhi + 1 < LB) {
          ++jhi;
          unsigned new_endj = xd_max(xd_min(jhi + 1, LB), endj);
          /* no guard here (the match branch has one): when endj == j this wipes the DPM[i+1][j+1]
           * just stored in Mrow[j] - reproduced, results depend on it; counted for the tests */
          if (endj + 1 <= new_endj && endj == j) ++g_xd_wipes;
          for (unsigned j2 = endj + 1; j2 <= new_endj; ++j2) { Mrow[j2 - 1] = XD_NEG; Drow[j2] = XD_NEG; }
          endj = new_endj;
        }
      }
      TBrow[j] = bits;
    }
    if (jhi < LB) {                                         /* end of Drow :645-666 */
      const unsigned j1 = jhi + 1;
      TBrow[j1] = 0;
      float md = M0 + Open;
      Drow[j1] += Ext;
      if (md >= Drow[j1]) { Drow[j1] = md; TBrow[j1] = XD_MD; }
    }
    if (next_jlo == UINT32_MAX) break;
    prev_jlo = jlo; prev_jhi = jhi;
    jlo = next_jlo; jhi = next_jhi;
    if (jlo >= LB) jlo = LB - 1;
    if (jhi >= LB) jhi = LB - 1;
    if (jlo == prev_jlo) { M0 = XD_NEG; Drow[jlo] = XD_NEG; }
    else M0 = Mrow[jlo - 1];
  }
  if (Best <= 0.0f) { *Leni = 0; *Lenj = 0; path[0] = 0; return 0.0f; }   /* :712-718 */

  /* XDropFwdTraceBackBitMem :271-342 */
  unsigned i = Besti, j = Bestj, n = 0;
  char State = 'M';
  for (;;) {
    path[n++] = State;
    if (i == 0 && j == 0) break;
    char Next;
    if (State == 'M') {
      byte c = TB[(size_t)i * W + j];
      Next = (c & XD_DM) ? 'D' : (c & XD_IM) ? 'I' : 'M';
      --i; --j;
    } else if (State == 'D') {
      byte c = TB[(size_t)i * W + j + 1];
      Next = (c & XD_MD) ? 'M' : 'D';
      --i;
    } else {
      byte c = TB[(size_t)(i + 1) * W + j];
      Next = (c & XD_MI) ? 'M' : 'I';
      --j;
    }
    State = Next;
  }
  for (unsigned k = 0; k < n / 2; ++k) { char t = path[k]; path[k] = path[n - 1 - k]; path[n - 1 - k] = t; }
  path[n] = 0;
  *Leni = Besti + 1; *Lenj = Bestj + 1;
  return Best;
}

/* XDropBwdFastMem xdropbwdmem.cpp:23-70: reverse both, extend forward, reverse the path */
static float xd_bwd(XdMem *mem, const float (*Sub)[256], float Open, float Ext,
                    const byte *A, unsigned LA, const byte *B, unsigned LB, float X,
                    unsigned *Leni, unsigned *Lenj, char *path)
{
  xd_alloc(mem, LA, LB);
  for (unsigned i = 0; i < LA; ++i) mem->reva[i] = A[LA - 1 - i];
  for (unsigned i = 0; i < LB; ++i) mem->revb[i] = B[LB - 1 - i];
  /* the reversed copies live in mem; xd_fwd's xd_alloc cannot move them (same sizes) */
  float Score = xd_fwd(mem, Sub, Open, Ext, mem->reva, LA, mem->revb, LB, X, Leni, Lenj, path);
  if (Score <= 0.0f) return Score;
  size_t n = strlen(path);
  for (size_t k = 0; k < n / 2; ++k) { char t = path[k]; path[k] = path[n - 1 - k]; path[n - 1 - k] = t; }
  return Score;
}

static unsigned xd_subl(unsigned L)                        /* GetSubL xdropfwdsplit.cpp:15-22 */
{
  if (L <= XD_MAXL) return L;
  if (L < 2 * XD_MAXL) return L / 2;
  return XD_MAXL;
}

/* XDropFwdSplit xdropfwdsplit.cpp:24-91 */
static float xd_fwd_split(XdMem *mem, const float (*Sub)[256], float Open, float Ext,
                          const byte *A, unsigned LA, const byte *B, unsigned LB, float X,
                          unsigned *Leni, unsigned *Lenj, char *path, char *tmp)
{
  *Leni = 0; *Lenj = 0; path[0] = 0;
  size_t n = 0;
  float Sum = 0.0f;
  for (;;) {
    if (*Leni == LA || *Lenj == LB) break;
    unsigned SubLA = xd_subl(LA - *Leni), SubLB = xd_subl(LB - *Lenj), si, sj;
    float Score = xd_fwd(mem, Sub, Open, Ext, A + *Leni, SubLA, B + *Lenj, SubLB, X, &si, &sj, tmp);
    if (Score == 0.0f) break;
    Sum += Score; *Leni += si; *Lenj += sj;
    size_t k = strlen(tmp);
    memcpy(path + n, tmp, k + 1); n += k;
    if (si < SubLA && sj < SubLB) break;
  }
  return Sum;
}

/* XDropBwdSplit xdropbwdsplit.cpp:15-79 */
static float xd_bwd_split(XdMem *mem, const float (*Sub)[256], float Open, float Ext,
                          const byte *A, unsigned LA, const byte *B, unsigned LB, float X,
                          unsigned *Leni, unsigned *Lenj, char *path, char *tmp)
{
  *Leni = 0; *Lenj = 0; path[0] = 0;
  size_t n = 0;
  float Sum = 0.0f;
  unsigned DoneA = 0, DoneB = 0;
  for (;;) {
    if (DoneA == LA || DoneB == LB) break;
    unsigned SubLA = xd_subl(LA - DoneA), SubLB = xd_subl(LB - DoneB), si, sj;
    const byte *SubA = A + LA - DoneA - SubLA, *SubB = B + LB - DoneB - SubLB;
    float Score = xd_bwd(mem, Sub, Open, Ext, SubA, SubLA, SubB, SubLB, X, &si, &sj, tmp);
    if (Score == 0.0f) break;
    Sum += Score; *Leni += si; *Lenj += sj;
    size_t k = strlen(tmp);                                 /* PrependPath */
    memmove(path + k, path, n + 1); memcpy(path, tmp, k); n += k;
    if (si < SubLA && sj < SubLB) break;
    DoneA += si; DoneB += sj;
  }
  return Sum;
}

static const float (*xd_subst(const ugs_xdrop_params *p, float (*custom)[256]))[256]
{
  init_tables();
  if (!p->is_nucleo) return g_subst_aa;
  if (p->match == 1.0f && p->mismatch == -2.0f) return g_subst_nt_default;
  fill_subst_nt(custom, p->match, p->mismatch);
  return custom;
}

void orc_xdrop_params_init(ugs_xdrop_params *p, int is_nucleo)
{
  memset(p, 0, sizeof *p);
  p->is_nucleo = is_nucleo;
  p->match = 1.0f; p->mismatch = -2.0f;                     /* o_defaults.inc -match/-mismatch */
  /* alnparams.cpp:362-369: -lopen/-lext defaults (o_defaults.inc:3-4: 10, 1) count as 'filled'
   * (opts.cpp:187-192), so the -5 aa branch at :373-376 is dead: -10/-1 for both alphabets */
  p->local_open = -10.0f;
  p->local_ext = -1.0f;
  p->xdrop = 32.0f;                                         /* o_defaults.inc:20 xdrop_g */
}

/* One job of the batched ABI.  path: NUL-terminated M/D/I, needs la+lb+2 bytes. */
int orc_xdrop_job(const ugs_xdrop_params *p, const char *a, uint32_t la, const char *b, uint32_t lb,
                  const ugs_xdrop_job *job, ugs_xdrop_hsp *hsp, char *path, uint64_t *cells)
{
  float (*custom)[256] = NULL;
  if (p->is_nucleo && !(p->match == 1.0f && p->mismatch == -2.0f)) custom = (float (*)[256])malloc(sizeof(float) * 65536);
  const float (*Sub)[256] = xd_subst(p, custom);
  const float Open = p->local_open, Ext = p->local_ext, X = p->xdrop;
  const byte *A = (const byte *)a, *B = (const byte *)b;
  XdMem mem; memset(&mem, 0, sizeof mem);
  memset(hsp, 0, sizeof *hsp);
  path[0] = 0;
  int rc = 0;
  if (job->mode == UGS_XDROP_FWD || job->mode == UGS_XDROP_BWD) {
    if (la == 0 || lb == 0 || la > XD_MAXL || lb > XD_MAXL) { rc = -1; goto done; }
    unsigned li, lj;
    float sc = job->mode == UGS_XDROP_FWD ? xd_fwd(&mem, Sub, Open, Ext, A, la, B, lb, X, &li, &lj, path)
                                          : xd_bwd(&mem, Sub, Open, Ext, A, la, B, lb, X, &li, &lj, path);
    hsp->score = sc; hsp->leni = li; hsp->lenj = lj;
    if (job->mode == UGS_XDROP_BWD) { hsp->loi = la - li; hsp->loj = lb - lj; }
    goto done;
  }
  /* XDropAlignMemMaxL2 xdropalignmem.cpp:26-214 */
  {
    const unsigned AncLoi = job->anc_loi, AncLoj = job->anc_loj, AncLen = job->anc_len;
    if (AncLen <= 1) goto done;                             /* :44-49 score 0, empty path */
    if (!(AncLoi < la && AncLoj < lb && AncLoi + AncLen <= la && AncLoj + AncLen <= lb)) { rc = -1; goto done; }
    const unsigned AncHii = AncLoi + AncLen - 1, AncHij = AncLoj + AncLen - 1;
    const unsigned FwdLA = la - AncHii, FwdLB = lb - AncHij;
    char *bp = (char *)malloc((size_t)la + lb + 4), *fp = (char *)malloc((size_t)la + lb + 4),
         *tmp = (char *)malloc((size_t)la + lb + 4);
    unsigned bi, bj, fi, fj;
    float BwdScore, FwdScore;
    if (AncLoi > XD_MAXL || AncLoj > XD_MAXL)
      BwdScore = xd_bwd_split(&mem, Sub, Open, Ext, A, AncLoi + 1, B, AncLoj + 1, X, &bi, &bj, bp, tmp);
    else
      BwdScore = xd_bwd(&mem, Sub, Open, Ext, A, AncLoi + 1, B, AncLoj + 1, X, &bi, &bj, bp);
    if (FwdLA > XD_MAXL || FwdLB > XD_MAXL)
      FwdScore = xd_fwd_split(&mem, Sub, Open, Ext, A + AncHii, FwdLA, B + AncHij, FwdLB, X, &fi, &fj, fp, tmp);
    else
      FwdScore = xd_fwd(&mem, Sub, Open, Ext, A + AncHii, FwdLA, B + AncHij, FwdLB, X, &fi, &fj, fp);
    size_t n = strlen(bp);
    memcpy(path, bp, n);
    memset(path + n, 'M', AncLen - 2); n += AncLen - 2;     /* :150 first & last anchor columns are in the x-drop paths */
    strcpy(path + n, fp);
    float AncScore = 0.0f;
    for (unsigned k = 0; k < AncLen; ++k) AncScore += Sub[A[AncLoi + k]][B[AncLoj + k]];
    float Dupe = Sub[A[AncLoi]][B[AncLoj]] + Sub[A[AncHii]][B[AncHij]];
    hsp->score = BwdScore + FwdScore + AncScore - Dupe;     /* :176 */
    hsp->loi = AncLoi + 1 - bi; hsp->loj = AncLoj + 1 - bj;
    hsp->leni = bi + fi + AncLen - 2; hsp->lenj = bj + fj + AncLen - 2;
    free(bp); free(fp); free(tmp);
  }
done:
  if (cells) *cells = mem.cells;
  xd_free(&mem);
  free(custom);
  return rc;
}
