// ugs_writers.cpp - host-side text writers beyond blast6/uc: -userout with
// -userfields, the -output_no_hits records, FASTA records, and HitMgr's hit-count rules.
// Pure formatting of device results (ugs_hit + run-length path + the two sequences); no search logic.
//
// Reference: OutputSink::OutputUser / OutputUserNoHits userout.cpp:47-352 (field list userfields.h:5-77),
// AlignResult getters arscorer.cpp / alignresult.h:97-240, OutputBlast6NoHits blast6out.cpp:82-103,
// SeqToFasta seqdb.cpp:62-90, HitMgr::GetHitCount hitmgr.cpp:366-393.
#include "ugs_dev.h"

#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

namespace {

// alpha2.cpp:220-300 identity classes, on upper- or lower-case letters
struct MatchTables {
  bool nt[256][256], aa[256][256];
  bool pos_nt[256][256], pos_aa[256][256];      // substitution score > 0 (setnucmx.cpp:11-99, blosum62.cpp)
  unsigned char comp[256];                      // alpha.cpp:3005-3265 g_CharToCompChar
  MatchTables()
  {
    int nucbit[256], iupac[256];
    memset(nucbit, 0, sizeof nucbit); memset(iupac, 0, sizeof iupac);
    nucbit['A'] = nucbit['a'] = 1; nucbit['C'] = nucbit['c'] = 2; nucbit['G'] = nucbit['g'] = 4;
    nucbit['T'] = nucbit['t'] = 8; nucbit['U'] = nucbit['u'] = 8;
    for (int c = 0; c < 256; ++c) iupac[c] = nucbit[c];
    const char *codes = "MRWSYKVHDBXN";
    const char *sets[] = {"AC", "AG", "AT", "CG", "CT", "GT", "ACG", "ACT", "AGT", "CGT", "GATC", "GATC"};
    for (int q = 0; codes[q]; ++q) {
      int b = 0;
      for (const char *s = sets[q]; *s; ++s) b |= nucbit[(unsigned char)*s];
      iupac[(unsigned char)codes[q]] = b; iupac[(unsigned char)tolower(codes[q])] = b;
    }
    for (int i = 0; i < 256; ++i) for (int j = 0; j < 256; ++j) {
      const bool ai = isalpha(i) != 0, aj = isalpha(j) != 0;
      if (!ai || !aj) { const bool g = (i == '-' || i == '.') && (j == '-' || j == '.'); nt[i][j] = aa[i][j] = g; continue; }
      if (toupper(i) == toupper(j)) { nt[i][j] = aa[i][j] = true; continue; }
      aa[i][j] = toupper(i) == 'X' || toupper(j) == 'X';
      nt[i][j] = (nucbit[i] & iupac[j]) || (nucbit[j] & iupac[i]);
    }
    aa['B']['N'] = aa['N']['B'] = aa['B']['D'] = aa['D']['B'] = true;
    aa['Z']['Q'] = aa['Q']['Z'] = aa['Z']['E'] = aa['E']['Z'] = true;
    memset(pos_nt, 0, sizeof pos_nt); memset(pos_aa, 0, sizeof pos_aa);
    for (int i = 0; i < 256; ++i) for (int j = 0; j < 256; ++j) {
      if (nucbit[i] && nucbit[j] && nucbit[i] == nucbit[j]) pos_nt[i][j] = true;      // default -match 1 > 0
      const char *pi = isalpha(i) ? strchr(UGS_B62_ORDER, toupper(i)) : nullptr, *pj = isalpha(j) ? strchr(UGS_B62_ORDER, toupper(j)) : nullptr;
      if (pi && pj && *pi && *pj) pos_aa[i][j] = UGS_B62[pi - UGS_B62_ORDER][pj - UGS_B62_ORDER] > 0;
    }
    pos_aa['*']['*'] = true;
    memset(comp, '?', sizeof comp);
    const char *from = "ABCDGHKMNRSTUVWXY", *to = "TVGHCDMKNYSAABWXR";
    for (int k = 0; from[k]; ++k) {
      comp[(unsigned char)from[k]] = (unsigned char)to[k];
      if (from[k] != 'U') comp[(unsigned char)tolower(from[k])] = (unsigned char)tolower(to[k]);
    }
  }
};
const MatchTables &tables() { static const MatchTables T; return T; }

enum Field {
  F_query, F_target, F_clusternr, F_evalue, F_id, F_fractid, F_dist, F_mid, F_pctpv, F_pctgaps, F_pairs, F_gaps, F_allgaps,
  F_qlo, F_qhi, F_tlo, F_thi, F_qlot, F_qhit, F_qunt, F_tlot, F_thit, F_tunt, F_pv, F_ql, F_tl, F_qs, F_ts, F_alnlen,
  F_opens, F_exts, F_raw, F_bits, F_aln, F_caln, F_qseq, F_tseq, F_qseg, F_tseg, F_qstrand, F_tstrand, F_qrow, F_trow,
  F_qrowdots, F_trowdots, F_qframe, F_tframe, F_mism, F_ids, F_qcov, F_tcov, F_diffs, F_diffsa, F_editdiffs,
  F_qlor, F_qhir, F_tlor, F_thir, F_orflo, F_orfhi, F_orfframe, F_COUNT
};
const char *const FIELD_NAMES[F_COUNT] = {
  "query", "target", "clusternr", "evalue", "id", "fractid", "dist", "mid", "pctpv", "pctgaps", "pairs", "gaps", "allgaps",
  "qlo", "qhi", "tlo", "thi", "qlot", "qhit", "qunt", "tlot", "thit", "tunt", "pv", "ql", "tl", "qs", "ts", "alnlen",
  "opens", "exts", "raw", "bits", "aln", "caln", "qseq", "tseq", "qseg", "tseg", "qstrand", "tstrand", "qrow", "trow",
  "qrowdots", "trowdots", "qframe", "tframe", "mism", "ids", "qcov", "tcov", "diffs", "diffsa", "editdiffs",
  "qlor", "qhir", "tlor", "thir", "orflo", "orfhi", "orfframe"};

// SetUserFieldIndexes userout.cpp:36-52
int parse_fields(const char *fields, std::vector<int> &out)
{
  out.clear();
  if (!fields || !*fields) { ugs_set_error("Invalid user fields ''"); return UGS_E_ARG; }
  std::string s(fields);
  size_t pos = 0;
  for (;;) {
    const size_t e = s.find('+', pos);
    const std::string name = s.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
    int f = -1;
    for (int k = 0; k < F_COUNT; ++k) if (name == FIELD_NAMES[k]) f = k;
    if (f < 0) { ugs_set_error("Invalid or unsupported user field name '%s'", name.c_str()); return UGS_E_ARG; }
    out.push_back(f);
    if (e == std::string::npos) break;
    pos = e + 1;
  }
  return UGS_OK;
}

void app(std::string &o, const char *fmt, ...)
{
  char tmp[128];
  va_list ap; va_start(ap, fmt);
  const int n = vsnprintf(tmp, sizeof tmp, fmt, ap);
  va_end(ap);
  if (n > 0) o.append(tmp, (size_t)(n < (int)sizeof tmp ? n : (int)sizeof tmp - 1));
}

double ratio(double x, double y) { return y == 0 ? 0.0 : x / y; }      // GetRatio myutils.h:234

int finish(const std::string &o, char *buf, int cap)
{
  if (buf && cap > 0) {
    const size_t n = o.size() < (size_t)cap - 1 ? o.size() : (size_t)cap - 1;
    memcpy(buf, o.data(), n); buf[n] = 0;
  }
  return (int)o.size();
}

}  // namespace

extern "C" int ugs_userfields_check(const char *fields)
{
  std::vector<int> f;
  return parse_fields(fields, f);
}

static int format_userout_impl(const ugs_params *lp, const ugs_hit *h, const uint32_t *cigar_pool, int is_nucleo, const char *fields,
                               const char *qlabel, const char *tlabel, const char *qseq, uint32_t ql,
                               const char *tseq, uint32_t tl, char *buf, int cap)
{
  // usearch_local hits (UGS_HIT_LOCAL): the HSP replaces the whole-sequence "HSP" of a global alignment in the coordinate,
  // segment and coverage getters (arscorer.cpp:122-154,688-806), and evalue / raw / bits are real (arscorer.cpp:69-120)
  const bool local = h && (h->flags & UGS_HIT_LOCAL);
  if (local && !lp) { ugs_set_error("local hits need the search parameters (ugs_format_userout_local)"); return UGS_E_ARG; }
  double lE = -1.0, lBits = 0.0;
  if (local) ugs_local_evalue(lp, (double)h->raw_score, h->ql, &lE, &lBits);
  std::vector<int> fs;
  const int rc = parse_fields(fields, fs);
  if (rc != UGS_OK) return rc;
  if (!qlabel || !qseq) { ugs_set_error("null argument"); return UGS_E_ARG; }
  std::string o;
  if (!h) {                                                  // OutputUserNoHits userout.cpp:47-105
    for (size_t i = 0; i < fs.size(); ++i) {
      if (i) o.push_back('\t');
      switch (fs[i]) {
        case F_query: o += qlabel; break;
        case F_ql: app(o, "%u", ql); break;
        case F_clusternr: app(o, "%u", 0xffffffffu); break;   // HM->m_QueryClusterIndex is UINT_MAX outside clustering
        case F_qseq: o.append(qseq, ql); break;
        case F_dist: case F_allgaps: case F_qlot: case F_qhit: case F_qunt: case F_tlot: case F_thit: case F_tunt: case F_qseg: case F_tseg:
        case F_qrowdots: case F_trowdots: case F_editdiffs: case F_orflo: case F_orfhi: case F_orfframe:
          ugs_set_error("Invalid user field index %s (-output_no_hits)", FIELD_NAMES[fs[i]]); return UGS_E_ARG;   // the reference dies on these
        default: o.push_back('*');
      }
    }
    o.push_back('\n');
    return finish(o, buf, cap);
  }
  if (!tlabel || !tseq || !cigar_pool) { ugs_set_error("null argument"); return UGS_E_ARG; }
  if (ql != h->ql || tl != h->tl) { ugs_set_error("sequence lengths do not match the hit record"); return UGS_E_ARG; }
  const MatchTables &T = tables();
  // the query as it was aligned (reverse-complemented for a minus-strand hit, seqinfo.cpp:292-323)
  std::string Q(qseq, ql);
  if (h->strand) for (uint32_t k = 0; k < ql; ++k) { const unsigned char c = (unsigned char)qseq[ql - 1 - k]; const unsigned char cc = T.comp[c]; Q[k] = (char)(cc == '?' ? c : cc); }
  std::string path;
  for (uint32_t k = 0; k < h->cigar_len; ++k) { const uint32_t r = cigar_pool[h->cigar_off + k]; path.append(r >> 2, "MDI"[r & 3]); }
  // AlignResult::FillLo arscorer.cpp:201-296 (only what the record does not already hold)
  const size_t cols = path.size();
  size_t firstM = path.find('M'), lastM = path.rfind('M');
  if (firstM == std::string::npos) { ugs_set_error("path has no match column"); return UGS_E_ARG; }
  const uint32_t qlo = h->qlo, tlo = h->tlo;                  // m_FirstMQPos / m_FirstMTPos
  const uint32_t aln = (uint32_t)(lastM - firstM + 1), term = (uint32_t)cols - aln;
  uint32_t diffsa = 0, pv = 0, opens = 0, exts = 0;
  std::string qrow, trow, qdots, tdots;
  {
    const bool (*Mx)[256] = is_nucleo ? T.nt : T.aa;
    const bool (*Pos)[256] = is_nucleo ? T.pos_nt : T.pos_aa;
    uint32_t qp = qlo, tp = tlo;
    char last = 'M';
    for (size_t c = firstM; c <= lastM; ++c) {
      const char op = path[c];
      const unsigned char qc = (op == 'M' || op == 'D') ? (unsigned char)Q[qp] : 0, tc = (op == 'M' || op == 'I') ? (unsigned char)tseq[tp] : 0;
      const char qu = qc ? (char)toupper(qc) : '-', tu = tc ? (char)toupper(tc) : '-';
      if (op == 'M') { if (qu != tu) ++diffsa; if (Pos[qc][tc]) ++pv; }
      if (op != 'M') { if (last == 'M') ++opens; else ++exts; }
      last = op;
      qrow.push_back(qu); trow.push_back(tu);
      qdots.push_back(qc ? (Mx[(unsigned char)qu][(unsigned char)tu] ? '.' : qu) : '-');
      tdots.push_back(tc ? (Mx[(unsigned char)qu][(unsigned char)tu] ? '.' : tu) : '-');
      if (qc) ++qp;
      if (tc) ++tp;
    }
  }
  const double fract = aln == 0 ? 0.0 : (double)h->ids / (double)aln;
  const uint32_t pairs = h->ids + h->mism;
  for (size_t i = 0; i < fs.size(); ++i) {
    if (i) o.push_back('\t');
    switch (fs[i]) {
      case F_query: o += qlabel; break;
      case F_target: o += tlabel; break;
      case F_clusternr: app(o, "%u", h->target); break;
      case F_evalue: app(o, "%.3g", lE); break;                // global: GetEvalue() = -1 (arscorer.cpp:69-73)
      case F_id: app(o, "%.1f", 100.0 * fract); break;
      case F_fractid: app(o, "%.4f", fract); break;
      case F_dist: app(o, "%.4f", 1.0 - fract); break;
      case F_mid: app(o, "%.1f", 100.0 * (h->ids == 0 ? 0.0 : (double)h->ids / (double)pairs)); break;
      case F_pctpv: app(o, "%.1f", 100.0 * ratio(pv, aln)); break;
      case F_pctgaps: app(o, "%.1f", 100.0 * ratio(h->gaps_int, aln)); break;
      case F_pairs: app(o, "%u", pairs); break;
      case F_gaps: app(o, "%u", h->gaps_int); break;
      case F_allgaps: app(o, "%u", h->gaps_int + term); break;
      // global HSP = whole sequences (alignresult.cpp:206-211); local: GetIQLo1/GetIQHi1 on the plus strand (arscorer.cpp:688-750)
      case F_qlo: app(o, "%u", !local ? 1u : (h->strand ? ql - h->qhi : h->qlo + 1)); break;
      case F_qhi: app(o, "%u", !local ? ql : (h->strand ? ql - h->qlo : h->qhi + 1)); break;
      case F_tlo: app(o, "%u", !local ? 1u : h->tlo + 1); break;
      case F_thi: app(o, "%u", !local ? tl : h->thi + 1); break;
      case F_qlor: app(o, "%u", !local ? 0u : h->qlo); break;
      case F_tlor: app(o, "%u", !local ? 0u : h->tlo); break;
      case F_qhir: app(o, "%u", !local ? ql - 1 : h->qhi); break;
      case F_thir: app(o, "%u", !local ? tl - 1 : h->thi); break;
      case F_qlot: app(o, "%u", h->qlo); break;
      case F_qhit: app(o, "%u", h->qhi); break;
      case F_qunt: app(o, "%u", ql - h->qhi - 1); break;
      case F_tlot: app(o, "%u", h->tlo); break;
      case F_thit: app(o, "%u", h->thi); break;
      case F_tunt: app(o, "%u", tl - h->thi - 1); break;
      case F_pv: app(o, "%u", pv); break;
      case F_ql: app(o, "%u", ql); break;
      case F_tl: app(o, "%u", tl); break;
      case F_qs: app(o, "%u", !local ? ql : h->qhi - h->qlo + 1); break;      // GetQuerySegLength: m_HSP.Leni
      case F_ts: app(o, "%u", !local ? tl : h->thi - h->tlo + 1); break;
      case F_alnlen: app(o, "%u", aln); break;
      case F_opens: app(o, "%u", opens); break;
      case F_exts: app(o, "%u", exts); break;
      case F_raw: app(o, "%.0f", local ? (double)h->raw_score : 0.0); break;   // global: 0 (arscorer.cpp:87-91,105-109)
      case F_bits: app(o, "%.0f", local ? lBits : 0.0); break;
      case F_aln: o += path; break;
      case F_caln:
        for (uint32_t k = 0; k < h->cigar_len; ++k) { const uint32_t r = cigar_pool[h->cigar_off + k]; if ((r >> 2) == 1) o.push_back("MDI"[r & 3]); else app(o, "%u%c", r >> 2, "MDI"[r & 3]); }
        break;
      case F_qseq: o += Q; break;
      case F_tseq: o.append(tseq, tl); break;
      // the reference prints segment-LENGTH (= whole sequence for a global hit) letters starting at the first
      // aligned position, i.e. it reads past the end when the alignment has a leading gap; here: up to the end
      case F_qseg: if (local) o.append(Q, h->qlo, h->qhi - h->qlo + 1); else o.append(Q, h->qlo, std::string::npos); break;
      case F_tseg: o.append(tseq + h->tlo, local ? h->thi - h->tlo + 1 : tl - h->tlo); break;
      case F_qstrand: o.push_back(!is_nucleo ? '.' : (h->strand ? '-' : '+')); break;
      case F_tstrand: o.push_back(!is_nucleo ? '.' : '+'); break;
      case F_qrow: o += qrow; break;
      case F_trow: o += trow; break;
      case F_qrowdots: o += qdots; break;
      case F_trowdots: o += tdots; break;
      case F_qframe: case F_tframe: case F_orfframe: app(o, "%+d", 0); break;
      case F_orflo: case F_orfhi: app(o, "%u", 0u); break;
      case F_mism: app(o, "%u", h->mism); break;
      case F_ids: app(o, "%u", h->ids); break;
      case F_qcov: app(o, "%.0f", 100.0 * (local ? (double)(h->qhi - h->qlo + 1) / (double)ql : (double)(h->qhi - h->qlo + 1) / ql)); break;
      case F_tcov: app(o, "%.0f", 100.0 * (local ? (double)(h->thi - h->tlo + 1) / (double)tl : (double)pairs / (double)tl)); break;
      case F_diffs: app(o, "%u", h->mism + h->gaps_int); break;
      case F_diffsa: app(o, "%u", diffsa); break;
      case F_editdiffs: app(o, "%u", h->mism + h->gaps_int + term); break;
    }
  }
  o.push_back('\n');
  return finish(o, buf, cap);
}

extern "C" int ugs_format_userout(const ugs_hit *h, const uint32_t *cigar_pool, int is_nucleo, const char *fields,
                                  const char *qlabel, const char *tlabel, const char *qseq, uint32_t ql,
                                  const char *tseq, uint32_t tl, char *buf, int cap)
{
  return format_userout_impl(nullptr, h, cigar_pool, is_nucleo, fields, qlabel, tlabel, qseq, ql, tseq, tl, buf, cap);
}

// the same for usearch_local hits: p supplies the Karlin-Altschul constants for evalue / bits
extern "C" int ugs_format_userout_local(const ugs_params *p, const ugs_hit *h, const uint32_t *cigar_pool, const char *fields,
                                        const char *qlabel, const char *tlabel, const char *qseq, uint32_t ql,
                                        const char *tseq, uint32_t tl, char *buf, int cap)
{
  if (!p) { ugs_set_error("null argument"); return UGS_E_ARG; }
  return format_userout_impl(p, h, cigar_pool, p->is_nucleo, fields, qlabel, tlabel, qseq, ql, tseq, tl, buf, cap);
}

// -trimout: OutputSink::OutputTrim outputsink.cpp:401-415 over AlignResult::GetTrimInfo arscorer.cpp:933-971: the query as
// aligned without the letters that hang over the target's ends (leading / trailing D runs of the path), label
// ":lo-hi" (1-based).  The reference copies positions [QLo, QHi) - the last kept letter is dropped - and so does this.
extern "C" int ugs_format_trimout(const ugs_hit *h, const uint32_t *cigar_pool, const char *qlabel, const char *qseq, uint32_t ql, char *buf, int cap)
{
  if (!h || !cigar_pool || !qlabel || !qseq) { ugs_set_error("null argument"); return UGS_E_ARG; }
  if (ql != h->ql) { ugs_set_error("sequence length does not match the hit record"); return UGS_E_ARG; }
  const MatchTables &T = tables();
  std::string Q(qseq, ql);
  if (h->strand) for (uint32_t k = 0; k < ql; ++k) { const unsigned char c = (unsigned char)qseq[ql - 1 - k]; const unsigned char cc = T.comp[c]; Q[k] = (char)(cc == '?' ? c : cc); }
  uint32_t QLo = 0, QHi = ql ? ql - 1 : 0;
  if (ql && h->cigar_len) {
    const uint32_t r0 = cigar_pool[h->cigar_off], rn = cigar_pool[h->cigar_off + h->cigar_len - 1];
    if ((r0 & 3) == 1) QLo = r0 >> 2;
    if ((rn & 3) == 1) { const uint32_t NewQHi = ql - (rn >> 2) - 1; if (NewQHi > QLo) QHi = NewQHi; }
  }
  std::string label(qlabel);
  app(label, ":%u-%u", QLo + 1, QHi + 1);
  const std::string seg = QHi > QLo ? Q.substr(QLo, QHi - QLo) : std::string();
  std::vector<char> rec(seg.size() + seg.size() / 80 + label.size() + 8);
  const int n = ugs_format_fasta(label.c_str(), seg.data(), (uint32_t)seg.size(), rec.data(), (int)rec.size());
  if (n < 0) return n;
  if (buf && cap > 0) { const int m = n < cap - 1 ? n : cap - 1; memcpy(buf, rec.data(), (size_t)m); buf[m] = 0; }
  return n;
}

// OutputBlast6NoHits blast6out.cpp:82-103 (written only under -output_no_hits)
extern "C" int ugs_format_blast6_nohit(const char *qlabel, char *buf, int cap)
{
  return snprintf(buf, (size_t)cap, "%s\t*\t0\t0\t0\t0\t0\t0\t0\t0\t*\t0\n", qlabel);
}

// SeqToFasta seqdb.cpp:62-90: ">label", then rows of fasta_cols (80) letters; nothing for an empty sequence
extern "C" int ugs_format_fasta(const char *label, const char *seq, uint32_t len, char *buf, int cap)
{
  if (len == 0) { if (buf && cap > 0) buf[0] = 0; return 0; }
  std::string o;
  if (label) { o.push_back('>'); o += label; o.push_back('\n'); }
  for (uint32_t p = 0; p < len; p += 80) { o.append(seq + p, len - p < 80 ? len - p : 80); o.push_back('\n'); }
  return finish(o, buf, cap);
}

// HitMgr::GetHitCount hitmgr.cpp:366-393 on the (already HitMgr::Sort-ed) hits of one query:
// -maxhits caps, -top_hits_only keeps the leading hits whose score (float fractional identity) equals the top
// score, -top_hit_only reports exactly one hit and that one is HitMgr::GetTopHit (hitmgr.cpp:395-415,464-467):
// the best score, ties to the smallest target index - not the sort order's first.  *first receives the index
// of the first hit to report (0 unless -top_hit_only).
extern "C" uint32_t ugs_hits_to_report(const ugs_hit *hits, uint32_t n, uint32_t maxhits, int top_hit_only, int top_hits_only,
                                       uint32_t *first)
{
  if (first) *first = 0;
  if (n == 0) return 0;
  // AlignResult::GetScore arscorer.cpp:818-824: fractional identity; usearch_local hits: the raw score
  auto score = [](const ugs_hit &h) { return (h.flags & UGS_HIT_LOCAL) ? h.raw_score : (h.aln_len == 0 ? 0.0f : (float)((double)h.ids / (double)h.aln_len)); };
  uint32_t count = n;
  if (maxhits && count > maxhits) count = maxhits;
  if (top_hit_only) {
    uint32_t best = 0;
    for (uint32_t i = 1; i < n; ++i)
      if (score(hits[i]) > score(hits[best]) || (score(hits[i]) == score(hits[best]) && (hits[i].target < hits[best].target ||
          (hits[i].target == hits[best].target && (hits[i].flags >> UGS_HIT_ORDER_SHIFT) < (hits[best].flags >> UGS_HIT_ORDER_SHIFT))))) best = i;
    if (first) *first = best;
    return 1;
  }
  if (top_hits_only) {
    float top = 0.0f;
    for (uint32_t i = 0; i < n; ++i) if (score(hits[i]) > top) top = score(hits[i]);
    for (uint32_t i = 1; i < count; ++i) if (score(hits[i]) < top) return i;
  }
  return count;
}

// ---------------------------------------------------------------- OTU table (otutab sink)
// OTUTableSink::OnQueryDone otutabsink.cpp:31-58 (top hit -> OTU name from the target label, sample name from
// the query label, count += size= annotation), OTUTable::IncCount / ToTabbedFile otutab.cpp:548-557,247-313
// (rows and columns in order of first appearance), label rules label.cpp:27-44,152-226.
#include <map>

namespace {

void split_fields(const std::string &s, char sep, std::vector<std::string> &out)      // Split() myutils.cpp: empty fields dropped
{
  out.clear();
  std::string cur;
  for (char c : s) { if (c == sep) { if (!cur.empty()) out.push_back(cur); cur.clear(); } else cur.push_back(c); }
  if (!cur.empty()) out.push_back(cur);
}

std::string str_field(const std::string &label, const char *name_eq)                  // GetStrField label.cpp:27-44
{
  std::vector<std::string> f;
  split_fields(label, ';', f);
  const size_t n = strlen(name_eq);
  for (const std::string &x : f) if (x.compare(0, n, name_eq) == 0) return x.substr(n);
  return std::string();
}

unsigned size_from_label(const std::string &label, unsigned dflt)                     // GetSizeFromLabel label.cpp:152-161
{
  const char *p = strstr(label.c_str(), ";size=");
  return p ? (unsigned)atoi(p + 6) : dflt;
}

std::string acc_from_label(const std::string &label)                                  // GetAccFromLabel label.cpp:168-182
{
  std::string acc;
  for (char c : label) {
    if (c == ' ' || c == '|' || c == ';') { if (acc != "gi") return acc; }
    acc.push_back(c);
  }
  return acc;
}

std::string sample_from_label(const std::string &label)                               // GetSampleNameFromLabel label.cpp:204-226
{
  std::string s = str_field(label, "sample=");
  if (!s.empty()) return s;
  s = str_field(label, "barcodelabel=");
  if (!s.empty()) return s;
  for (char c : label) { if (!isalpha((unsigned char)c) && !isdigit((unsigned char)c) && c != '_') break; s.push_back(c); }
  return s;
}

}  // namespace

struct ugs_otutab {
  std::vector<std::string> otus, samples;
  std::map<std::string, unsigned> otu_ix, sample_ix;
  std::vector<std::vector<unsigned>> counts;          // [otu][sample]
  unsigned long long assigned = 0, total = 0;
};

extern "C" ugs_otutab *ugs_otutab_create(void) { return new ugs_otutab(); }
extern "C" void ugs_otutab_destroy(ugs_otutab *t) { delete t; }

extern "C" int ugs_otutab_add(ugs_otutab *t, const char *qlabel, const char *top_hit_tlabel, char *map_line, int cap)
{
  if (!t || !qlabel) { ugs_set_error("null argument"); return UGS_E_ARG; }
  const std::string ql(qlabel);
  const unsigned size = size_from_label(ql, 1);
  t->total += size;
  if (map_line && cap > 0) map_line[0] = 0;
  if (!top_hit_tlabel) return 0;
  const std::string tl(top_hit_tlabel);
  std::string otu = str_field(tl, "otu=");                                            // GetOTUNameFromLabel label.cpp:193-202
  if (otu.empty()) otu = acc_from_label(tl);
  if (otu.empty()) { ugs_set_error("Empty OTU name in label >%s", top_hit_tlabel); return UGS_E_ARG; }
  const std::string sample = sample_from_label(ql);
  t->assigned += size;
  unsigned oi, si;
  auto io = t->otu_ix.find(otu);
  if (io == t->otu_ix.end()) { oi = (unsigned)t->otus.size(); t->otus.push_back(otu); t->otu_ix[otu] = oi; t->counts.emplace_back(t->samples.size(), 0u); }
  else oi = io->second;
  auto is = t->sample_ix.find(sample);
  if (is == t->sample_ix.end()) { si = (unsigned)t->samples.size(); t->samples.push_back(sample); t->sample_ix[sample] = si; for (auto &row : t->counts) row.push_back(0u); }
  else si = is->second;
  t->counts[oi][si] += size;
  return snprintf(map_line, map_line ? (size_t)cap : 0, "%s\t%s\n", qlabel, otu.c_str());
}

extern "C" int ugs_otutab_write(const ugs_otutab *t, const char *path)
{
  if (!t || !path) { ugs_set_error("null argument"); return UGS_E_ARG; }
  FILE *f = fopen(path, "w");
  if (!f) { ugs_set_error("cannot create %s", path); return UGS_E_ARG; }
  fputs("#OTU ID", f);
  for (const std::string &s : t->samples) { fputc('\t', f); fputs(s.c_str(), f); }
  fputc('\n', f);
  for (size_t o = 0; o < t->otus.size(); ++o) {
    fputs(t->otus[o].c_str(), f);
    for (size_t s = 0; s < t->samples.size(); ++s) fprintf(f, "\t%u", t->counts[o][s]);
    fputc('\n', f);
  }
  fclose(f);
  return UGS_OK;
}

// OTUTable::ToJsonFile json.cpp:32-103 (-biomout): BIOM 1.0, sparse triples.  The comma after a triple is decided by the
// triple's own indexes, not by whether another one follows (so a table whose last cell is zero ends in ",") - reproduced.
extern "C" int ugs_otutab_write_biom(const ugs_otutab *t, const char *path)
{
  if (!t || !path) { ugs_set_error("null argument"); return UGS_E_ARG; }
  FILE *f = fopen(path, "w");
  if (!f) { ugs_set_error("cannot create %s", path); return UGS_E_ARG; }
  const unsigned no = (unsigned)t->otus.size(), nsam = (unsigned)t->samples.size();
  time_t now = time(nullptr);
  char ts[26];
  memcpy(ts, asctime(localtime(&now)), 24); ts[24] = 0;
  fprintf(f, "{\n\t\"id\":\"%s\",\n\t\"format\": \"Biological Observation Matrix 1.0\",\n\t\"format_url\": \"http://biom-format.org\",\n", path);
  fprintf(f, "\t\"generated_by\": \"usearch\",\n\t\"type\": \"OTU table\",\n\t\"date\": \"%s\",\n\t\"matrix_type\": \"sparse\",\n", ts);
  fprintf(f, "\t\"matrix_element_type\": \"float\",\n\t\"shape\": [%u,%u],\n\t\"rows\":[\n", no, nsam);
  for (unsigned o = 0; o < no; ++o) fprintf(f, "\t\t{\"id\":\"%s\", \"metadata\":null}%s\n", t->otus[o].c_str(), o + 1 != no ? "," : "");
  fprintf(f, "\t],\n\t\"columns\":[\n");
  for (unsigned k = 0; k < nsam; ++k) fprintf(f, "\t\t{\"id\":\"%s\", \"metadata\":null}%s\n", t->samples[k].c_str(), k + 1 != nsam ? "," : "");
  fprintf(f, "\t],\n\t\"data\": [\n");
  for (unsigned o = 0; o < no; ++o)
    for (unsigned k = 0; k < nsam; ++k) {
      const unsigned c = t->counts[o][k];
      if (c == 0) continue;
      fprintf(f, "\t\t[%u,%u,%u]%s\n", o, k, c, (o + 1 < no || k + 1 < nsam) ? "," : "");
    }
  fprintf(f, "\t]\n}\n");
  fclose(f);
  return UGS_OK;
}

extern "C" int ugs_otutab_totals(const ugs_otutab *t, uint64_t *assigned, uint64_t *total)
{
  if (!t) return UGS_E_ARG;
  if (assigned) *assigned = t->assigned;
  if (total) *total = t->total;
  return UGS_OK;
}

// ---------------------------------------------------------------- closed_ref sink
// ClosedRefSink::OnQueryDone closedrefsink.cpp:33-118: the top hit's target becomes (or is) a reference OTU in order of
// first use; -tabbedout gets one line per query.  (-dbotus / -dataotus, OnAllDone :120-164, are not built: the
// reference itself crashes when either is given - it keeps pointers into recycled SeqInfo objects - so there is
// nothing to pin them against.)
namespace {
}  // namespace

struct ugs_closedref {
  std::map<uint32_t, unsigned> target_to_otu;
  std::vector<std::string> ref_labels;
  std::vector<unsigned> total_size, members;
  unsigned long long assigned = 0, unassigned = 0;
};

extern "C" ugs_closedref *ugs_closedref_create(void) { return new ugs_closedref(); }
extern "C" void ugs_closedref_destroy(ugs_closedref *c) { delete c; }

extern "C" int ugs_closedref_add(ugs_closedref *c, const char *qlabel, const ugs_hit *hits, uint32_t n, const char *const *tlabels, char *line, int cap)
{
  if (!c || !qlabel || (n && (!hits || !tlabels))) { ugs_set_error("null argument"); return UGS_E_ARG; }
  if (n == 0) { ++c->unassigned; return snprintf(line, line ? (size_t)cap : 0, "%s\t*\t*\t*\t*\t*\n", qlabel); }
  const unsigned size = size_from_label(qlabel, 1);
  uint32_t top = 0;
  ugs_hits_to_report(hits, n, 0, 1, 0, &top);                                             // HitMgr::GetTopHit
  ++c->assigned;
  const uint32_t tt = hits[top].target;
  auto fid = [&](uint32_t i) { return (float)(hits[i].aln_len == 0 ? 0.0 : (double)hits[i].ids / (double)hits[i].aln_len); };   // HitMgr::GetFractId
  const double TopFractId = fid(0);
  unsigned otu;
  auto it = c->target_to_otu.find(tt);
  if (it == c->target_to_otu.end()) {
    otu = (unsigned)c->ref_labels.size();
    c->target_to_otu[tt] = otu;
    c->ref_labels.emplace_back(tlabels[top]);
    c->total_size.push_back(0); c->members.push_back(0);
  } else otu = it->second;
  c->total_size[otu] += size;
  const unsigned member = c->members[otu]++;
  unsigned ties = 0;
  std::string ties_str;
  if (n > 1)
    for (uint32_t i = 0; i < n; ++i) {
      if ((double)fid(i) < TopFractId) break;
      if (hits[i].target == tt) continue;
      if (ties > 0) ties_str += ",";
      ties_str += tlabels[i];
      ++ties;
    }
  std::string out = std::string(qlabel) + "\t" + std::to_string(otu) + "\t" + std::to_string(member) + "\t" + tlabels[top];
  char num[64];
  snprintf(num, sizeof num, "\t%.1f\tties=%u", TopFractId * 100.0, ties);
  out += num;
  if (ties > 0) out += ":" + ties_str;
  out += "\n";
  return snprintf(line, line ? (size_t)cap : 0, "%s", out.c_str());
}

extern "C" int ugs_closedref_totals(const ugs_closedref *c, uint64_t *assigned, uint64_t *unassigned, uint32_t *otus)
{
  if (!c) return UGS_E_ARG;
  if (assigned) *assigned = c->assigned;
  if (unassigned) *unassigned = c->unassigned;
  if (otus) *otus = (uint32_t)c->total_size.size();
  return UGS_OK;
}

// ---------------------------------------------------------------- -alnout
// OutputSink::OutputReport / OutputReportGlobal outputsink.cpp:243-258,338-356 (per-query hit table) and WriteAln
// alnout.cpp:41-171 (the alignment in rows of -rowlen 80 columns with position labels, annotation row
// arscorer.cpp:12-45,481-504 and the summary line), global-alignment semantics, no ORFs.
namespace {
unsigned ndig(unsigned n) { return n < 10 ? 1 : n < 100 ? 2 : n < 1000 ? 3 : n < 10000 ? 4 : n < 100000 ? 5 : n < 1000000 ? 6 : 10; }   // alnout.cpp:8-23

// Pos is the offset of the first letter, returns the offset of the last one (alnout.cpp:25-39)
unsigned advance_pos(unsigned pos, const char *row, unsigned n, bool *all_gaps)
{
  unsigned np = pos; bool got = false;
  for (unsigned i = 0; i < n; ++i) if (row[i] != '-') { if (got) ++np; else got = true; }
  *all_gaps = !got;
  return np;
}
}  // namespace

extern "C" int ugs_format_alnout_header(const ugs_hit *hits, uint32_t n, const char *qlabel, const char *const *tlabels, char *buf, int cap)
{
  if (n == 0) { if (buf && cap > 0) buf[0] = 0; return 0; }            // OutputReport prints nothing for a query without hits
  if (!hits || !qlabel || !tlabels) { ugs_set_error("null argument"); return UGS_E_ARG; }
  std::string o = "\nQuery >";
  o += qlabel; o += "\n %Id   TLen  Target\n";
  for (uint32_t i = 0; i < n; ++i) {
    const double pct = 100.0 * (hits[i].aln_len == 0 ? 0.0 : (double)hits[i].ids / (double)hits[i].aln_len);
    app(o, "%3.0f%%  %5u  ", pct, hits[i].tl);
    o += tlabels[i]; o.push_back('\n');
  }
  return finish(o, buf, cap);
}

// OutputReportLocal outputsink.cpp:260-298 (+ OutputReport :338-356): score, e-value, identity, query / target segment as
// lo-hi(unaligned tail) in plus-strand query coordinates (FormatSeg :57-62), strand for nucleotide queries
extern "C" int ugs_format_alnout_header_local(const ugs_params *p, const ugs_hit *hits, uint32_t n, const char *qlabel,
                                              const char *const *tlabels, char *buf, int cap)
{
  if (n == 0) { if (buf && cap > 0) buf[0] = 0; return 0; }
  if (!p || !hits || !qlabel || !tlabels) { ugs_set_error("null argument"); return UGS_E_ARG; }
  std::string o = "\nQuery >";
  o += qlabel; o += "\n Score     Evalue   %Id    QueryLo-Hi(Un)   TargetLo-Hi(Un)";
  if (p->is_nucleo) o += "  +";
  o += "  Target\n";
  for (uint32_t i = 0; i < n; ++i) {
    const ugs_hit &h = hits[i];
    double E = 0, bits = 0;
    ugs_local_evalue(p, (double)h.raw_score, h.ql, &E, &bits);
    const double pct = 100.0 * (h.aln_len == 0 ? 0.0 : (double)h.ids / (double)h.aln_len);
    const unsigned qlo = h.strand ? h.ql - h.qhi - 1 : h.qlo, qhi = h.strand ? h.ql - h.qlo - 1 : h.qhi;      // GetIQLo / GetIQHi arscorer.cpp:688-750
    char seg[64];
    app(o, "%6.0f  %9.1g  %3.0f%%", (double)h.raw_score, E, pct);
    snprintf(seg, sizeof seg, "%u-%u(%u)", qlo + 1, qhi + 1, h.ql - qhi - 1); app(o, "  %16s", seg);
    snprintf(seg, sizeof seg, "%u-%u(%u)", h.tlo + 1, h.thi + 1, h.tl - h.thi - 1); app(o, "  %16s", seg);
    if (p->is_nucleo) { o += "  "; o.push_back(h.strand ? '-' : '+'); }
    o += "  "; o += tlabels[i]; o.push_back('\n');
  }
  return finish(o, buf, cap);
}

static int format_alnout_hit_impl(const ugs_params *lp, const ugs_hit *h, const uint32_t *cigar_pool, int is_nucleo, const char *qlabel,
                                  const char *tlabel, const char *qseq, uint32_t ql, const char *tseq, uint32_t tl,
                                  char *buf, int cap)
{
  if (!h || !cigar_pool || !qlabel || !tlabel || !qseq || !tseq) { ugs_set_error("null argument"); return UGS_E_ARG; }
  if (ql != h->ql || tl != h->tl) { ugs_set_error("sequence lengths do not match the hit record"); return UGS_E_ARG; }
  const MatchTables &T = tables();
  std::string Q(qseq, ql);
  if (h->strand) for (uint32_t k = 0; k < ql; ++k) { const unsigned char c = (unsigned char)qseq[ql - 1 - k]; const unsigned char cc = T.comp[c]; Q[k] = (char)(cc == '?' ? c : cc); }
  std::string path;
  for (uint32_t k = 0; k < h->cigar_len; ++k) { const uint32_t r = cigar_pool[h->cigar_off + k]; path.append(r >> 2, "MDI"[r & 3]); }
  const size_t firstM = path.find('M'), lastM = path.rfind('M');
  if (firstM == std::string::npos) { ugs_set_error("path has no match column"); return UGS_E_ARG; }
  std::string qrow, trow, arow;
  {
    uint32_t qp = h->qlo, tp = h->tlo;
    for (size_t c = firstM; c <= lastM; ++c) {
      const char op = path[c];
      const unsigned char qc = (op == 'M' || op == 'D') ? (unsigned char)Q[qp] : 0, tc = (op == 'M' || op == 'I') ? (unsigned char)tseq[tp] : 0;
      qrow.push_back(qc ? (char)toupper(qc) : '-'); trow.push_back(tc ? (char)toupper(tc) : '-');
      char sym = ' ';
      if (op == 'M') {
        if (is_nucleo) {                                        // GetNucleoSym arscorer.cpp:27-36
          const bool acgtu_q = strchr("ACGTU", toupper(qc)) != nullptr, acgtu_t = strchr("ACGTU", toupper(tc)) != nullptr;
          if (toupper(qc) == toupper(tc) && acgtu_q && acgtu_t) sym = '|';
          else if (T.nt[qc][tc]) sym = '+';
        } else {                                                // GetAminoSym arscorer.cpp:12-25
          if (T.aa[qc][tc]) sym = '|';
          else {
            const char *pi = isalpha(qc) ? strchr(UGS_B62_ORDER, toupper(qc)) : nullptr, *pj = isalpha(tc) ? strchr(UGS_B62_ORDER, toupper(tc)) : nullptr;
            const int sc = (pi && pj && *pi && *pj) ? UGS_B62[pi - UGS_B62_ORDER][pj - UGS_B62_ORDER] : (qc == '*' && tc == '*' ? 1 : ((qc == '*' || tc == '*') ? -4 : 0));
            sym = sc >= 2 ? ':' : (sc > 0 ? '.' : ' ');
          }
        }
      }
      arow.push_back(sym);
      if (qc) ++qp;
      if (tc) ++tp;
    }
  }
  const unsigned aln = (unsigned)qrow.size();
  const unsigned mx = ql > tl ? ql : tl, w = ndig(mx);
  const char *unit = is_nucleo ? "nt" : "aa";
  const char qstrand = !is_nucleo ? '.' : (h->strand ? '-' : '+'), tstrand = !is_nucleo ? '.' : '+';
  const bool show_strand = qstrand != '.';
  std::string o = "\n";
  app(o, " Query %*u%s >", (int)w, ql, unit); o += qlabel; o.push_back('\n');
  app(o, "Target %*u%s >", (int)w, tl, unit); o += tlabel; o.push_back('\n');
  o.push_back('\n');
  const unsigned rowlen = 80;
  unsigned qpos = h->qlo, tpos = h->tlo;
  bool qgaps = false, tgaps = false;
  auto ipos_q = [&](unsigned pos) { return h->strand ? ql - pos - 1 : pos; };       // PosToIPosQ arscorer.cpp:598-645 (no ORF)
  for (unsigned from = 0; from < aln; from += rowlen) {
    const unsigned n = aln - from < rowlen ? aln - from : rowlen;
    const unsigned qfrom = ipos_q(qpos) + (qgaps ? 0 : 1), tfrom = tpos + (tgaps ? 0 : 1);     // PosToIPosQ1 / T1 with the PREVIOUS row's all-gaps flag
    qpos = advance_pos(qpos, qrow.c_str() + from, n, &qgaps);
    tpos = advance_pos(tpos, trow.c_str() + from, n, &tgaps);
    const unsigned qto = ipos_q(qpos) + (qgaps ? 0 : 1), tto = tpos + (tgaps ? 0 : 1);
    if (!qgaps) ++qpos;
    if (!tgaps) ++tpos;
    app(o, "Qry %*u", (int)w, qfrom); if (show_strand) { o.push_back(' '); o.push_back(qstrand); }
    o.push_back(' '); o.append(qrow, from, n); app(o, " %u\n", qto);
    o += "    "; o.append(w, ' '); if (show_strand) o += "  ";
    o.push_back(' '); o.append(arow, from, n); o.push_back('\n');
    app(o, "Tgt %*u", (int)w, tfrom); if (show_strand) { o.push_back(' '); o.push_back(tstrand); }
    o.push_back(' '); o.append(trow, from, n); app(o, " %u\n", tto);
    o.push_back('\n');
  }
  app(o, "%u cols, %u ids (%.1f%%), %u gaps (%.1f%%)", aln, h->ids, 100.0 * ratio(h->ids, aln), h->gaps_int, 100.0 * ratio(h->gaps_int, aln));
  if (lp && (h->flags & UGS_HIT_LOCAL)) {                              // alnout.cpp:151-163
    double E = 0, bits = 0;
    ugs_local_evalue(lp, (double)h->raw_score, h->ql, &E, &bits);
    app(o, ", score %.1f (%.1f bits), Evalue %.2g", (double)h->raw_score, bits, E);
  }
  o.push_back('\n');
  return finish(o, buf, cap);
}

extern "C" int ugs_format_alnout_hit(const ugs_hit *h, const uint32_t *cigar_pool, int is_nucleo, const char *qlabel,
                                     const char *tlabel, const char *qseq, uint32_t ql, const char *tseq, uint32_t tl,
                                     char *buf, int cap)
{
  return format_alnout_hit_impl(nullptr, h, cigar_pool, is_nucleo, qlabel, tlabel, qseq, ql, tseq, tl, buf, cap);
}

// WriteAln for a usearch_local hit: the same rows over the HSP, the summary line ends with score, bits and e-value
extern "C" int ugs_format_alnout_hit_local(const ugs_params *p, const ugs_hit *h, const uint32_t *cigar_pool, const char *qlabel,
                                           const char *tlabel, const char *qseq, uint32_t ql, const char *tseq, uint32_t tl,
                                           char *buf, int cap)
{
  if (!p) { ugs_set_error("null argument"); return UGS_E_ARG; }
  return format_alnout_hit_impl(p, h, cigar_pool, p->is_nucleo, qlabel, tlabel, qseq, ql, tseq, tl, buf, cap);
}

// ---------------------------------------------------------------- -fastapairs / -qsegout / -tsegout
// OutputSink::OutputFastaPairs outputsink.cpp:231-241, OutputQSeg / OutputTSeg :203-229 with RowToFasta :30-55
// (the aligned rows of AlignResult::GetQueryRow / GetTargetRow, arscorer.cpp:305-324,506-525).
namespace {
int aligned_rows(const ugs_hit *h, const uint32_t *cigar_pool, const char *qseq, uint32_t ql, const char *tseq, uint32_t tl,
                 std::string &qrow, std::string &trow)
{
  if (!h || !cigar_pool || !qseq || !tseq) { ugs_set_error("null argument"); return UGS_E_ARG; }
  if (ql != h->ql || tl != h->tl) { ugs_set_error("sequence lengths do not match the hit record"); return UGS_E_ARG; }
  const MatchTables &T = tables();
  std::string Q(qseq, ql);
  if (h->strand) for (uint32_t k = 0; k < ql; ++k) { const unsigned char c = (unsigned char)qseq[ql - 1 - k]; const unsigned char cc = T.comp[c]; Q[k] = (char)(cc == '?' ? c : cc); }
  std::string path;
  for (uint32_t k = 0; k < h->cigar_len; ++k) { const uint32_t r = cigar_pool[h->cigar_off + k]; path.append(r >> 2, "MDI"[r & 3]); }
  const size_t firstM = path.find('M'), lastM = path.rfind('M');
  if (firstM == std::string::npos) { ugs_set_error("path has no match column"); return UGS_E_ARG; }
  uint32_t qp = h->qlo, tp = h->tlo;
  qrow.clear(); trow.clear();
  for (size_t c = firstM; c <= lastM; ++c) {
    const char op = path[c];
    qrow.push_back((op == 'M' || op == 'D') ? (char)toupper((unsigned char)Q[qp++]) : '-');
    trow.push_back((op == 'M' || op == 'I') ? (char)toupper((unsigned char)tseq[tp++]) : '-');
  }
  return UGS_OK;
}

void row_to_fasta(std::string &o, const char *label, const std::string &row)     // RowToFasta outputsink.cpp:30-55
{
  o.push_back('>'); o += label;
  unsigned out = 0;
  for (char c : row) {
    if (c == '-' || c == '.') continue;
    if (out % 80 == 0) o.push_back('\n');
    o.push_back(c); ++out;
  }
  o.push_back('\n');
}
}  // namespace

extern "C" int ugs_format_fastapairs(const ugs_hit *h, const uint32_t *cigar_pool, const char *qlabel, const char *tlabel,
                                     const char *qseq, uint32_t ql, const char *tseq, uint32_t tl, char *buf, int cap)
{
  std::string qrow, trow;
  const int rc = aligned_rows(h, cigar_pool, qseq, ql, tseq, tl, qrow, trow);
  if (rc != UGS_OK) return rc;
  std::string o = ">"; o += qlabel; o.push_back('\n'); o += qrow; o += "\n>"; o += tlabel; o.push_back('\n'); o += trow; o += "\n\n";
  return finish(o, buf, cap);
}

// which = 0: -qsegout (the aligned part of the query), 1: -tsegout (of the target)
extern "C" int ugs_format_segout(const ugs_hit *h, const uint32_t *cigar_pool, int which, const char *qlabel, const char *tlabel,
                                 const char *qseq, uint32_t ql, const char *tseq, uint32_t tl, char *buf, int cap)
{
  std::string qrow, trow, o;
  const int rc = aligned_rows(h, cigar_pool, qseq, ql, tseq, tl, qrow, trow);
  if (rc != UGS_OK) return rc;
  if (which == 0) row_to_fasta(o, qlabel, qrow); else row_to_fasta(o, tlabel, trow);
  return finish(o, buf, cap);
}
