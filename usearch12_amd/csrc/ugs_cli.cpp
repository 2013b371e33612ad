// ugs_cli.cpp - host-side driver with the reference's command-line surface for the one command
// this repository accelerates:
//
//   ugs_cli -usearch_global q.fa -db db.fa|db.udb -id 0.97 -strand plus|both [-blast6out f] [-uc f]
//           [-maxaccepts n] [-maxrejects n] [-big n] [-device n] [-batch n]
//           [-wordlength n] [-stepwords n] [-bump n] [-hspw n] [-minhsp n] [-xdrop_nw x] [-band n] [-match x] [-mismatch x]
//   ugs_cli -makeudb_usearch db.fa -output db.udb [-dbtype nt|aa]       (makeudb.cpp:27-66; index built on the GPU)
//   ugs_cli -cluster_fast reads.fa -id 0.97 [-strand both] [-sort length|size] [-sizein] [-sizeout] [-minsize n] -uc c.uc -centroids c.fa   (clusterfast.cpp:37-138)
//   ugs_cli -usearch_local q.fa -db db.fa|db.udb -evalue 1e-6 [-id ..] -strand plus|both -blast6out f
//   ugs_cli -closed_ref reads.fa -db ref.fa -strand plus|both -tabbedout f
//   ugs_cli -otutab reads.fa -otus otus.fa|-zotus ..|-db .. [-otutabout f] [-mapout f] [+ the usearch_global outputs]
//           (cmd_otutab searchcmd.cpp:20-40: defaults -id 0.97 -maxaccepts 3 -maxrejects 32 -stepwords 0 -strand both)
//
// It stands where cmd_usearch_global -> Search() -> Thread() stand in the reference
// (searchcmd.cpp:6-9, search.cpp:51-141): load the DB, stream query batches through the C-ABI
// (include/ugs.h), write hits in query input order (== the reference at -threads 1).
// The mirror of the reference's object surface is deliberately thin: Searcher::Search(batch),
// HitMgr (hit grouping, done inside ugs_batch_fetch) and OutputSink (the text writers).
#include "../../include/ugs.h"
#include "../../include/ugs_comm.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include <algorithm>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <queue>
#include <thread>

struct SeqSet {                       // SeqDB (seqdb.h:29-52) flattened: labels + concatenated letters
  std::vector<std::string> labels;
  std::string letters;
  std::string quals;                  // FASTQ input only: one quality character per letter (same offsets)
  std::vector<uint64_t> offs{0};
  size_t size() const { return labels.size(); }
};

// FASTASeqSource::GetNextLo (fastaseqsource.cpp:25-124): label = everything after '>'; whitespace and
// gap characters are stripped from sequences, other non-alpha bytes are dropped; empty records skipped.
class FastaReader {
 public:
  explicit FastaReader(const char *path) : f_(fopen(path, "rb")) {
    if (!f_) { fprintf(stderr, "cannot open %s\n", path); exit(1); }
    have_ = next_line();
  }
  ~FastaReader() { if (f_) fclose(f_); }
  // append up to max_seqs records to out; returns number appended
  size_t read(SeqSet &out, size_t max_seqs) {
    size_t n = 0;
    while (n < max_seqs && have_) {
      if (line_.empty()) { have_ = next_line(); continue; }
      if (line_[0] == '@' && (fastq_ || out.size() == 0 || !out.quals.empty())) {
        // FASTQSeqSource::GetNextLo fastqseqsource.cpp:7-107: '@'label / letters / '+'... / qualities, one line each
        fastq_ = true;
        std::string label = line_.substr(1);
        if (!next_line()) { fprintf(stderr, "Unexpected end-of-file in FASTQ file\n"); exit(1); }
        for (unsigned char c : line_) if (!isalpha(c)) { fprintf(stderr, "Invalid sequence letter in FASTQ\n"); exit(1); }
        const std::string seq = line_;
        next_line();                                                   // '+' line: contents ignored
        if (!next_line() || line_.size() != seq.size()) { fprintf(stderr, "Bad FASTQ record: %zu bases, %zu quals\n", seq.size(), line_.size()); exit(1); }
        have_ = true;
        if (!seq.empty()) {
          out.letters += seq; out.quals += line_;
          out.labels.push_back(label); out.offs.push_back(out.letters.size()); ++n;
        }
        have_ = next_line();
        continue;
      }
      if (line_[0] != '>') { fprintf(stderr, "bad FASTA: expected '>'\n"); exit(1); }
      std::string label = line_.substr(1);
      const size_t start = out.letters.size();
      while ((have_ = next_line()) && !(line_.size() && line_[0] == '>'))
        {   // (almost every line is letters only: one check, one append)
          bool pure = true;
          for (unsigned char c : line_) pure &= (unsigned char)((c | 0x20) - 'a') < 26;
          if (pure) out.letters.append(line_);
          else for (unsigned char c : line_) if ((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z')) out.letters.push_back((char)c);
        }
      if (out.letters.size() == start) continue;                 // empty sequence: skipped (with a warning in the reference)
      out.labels.push_back(label); out.offs.push_back(out.letters.size()); ++n;
    }
    return n;
  }
 private:
  // one line without its end-of-line ('\r' dropped); reads the file in 4 MiB blocks and finds line ends with memchr
  bool next_line() {
    line_.clear();
    bool any = false;
    for (;;) {
      if (pos_ == len_) {
        len_ = fread(buf_.data(), 1, buf_.size(), f_);
        pos_ = 0;
        if (len_ == 0) break;
      }
      any = true;
      const char *b = buf_.data() + pos_;
      const char *nl = (const char *)memchr(b, '\n', len_ - pos_);
      const size_t n = nl ? (size_t)(nl - b) : len_ - pos_;
      line_.append(b, n);
      pos_ += n + (nl ? 1 : 0);
      if (nl) break;
    }
    if (!line_.empty() && line_.find('\r') != std::string::npos) {
      size_t w = 0;
      for (char c : line_) if (c != '\r') line_[w++] = c;
      line_.resize(w);
    }
    return any;
  }
  std::vector<char> buf_ = std::vector<char>(4u << 20);
  size_t pos_ = 0, len_ = 0;
  FILE *f_; std::string line_; bool have_; bool fastq_ = false;
};

// Whole-file FASTA parse by several threads (same rules as FastaReader): the file is cut at record starts into one piece per
// thread, every piece is parsed into its own SeqSet, the pieces are concatenated in order.  FASTQ and tiny files take the
// serial reader.
static void parse_fasta_piece(const char *p, const char *e, SeqSet &out)
{
  while (p < e) {
    const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
    const char *le = nl ? nl : e;
    if (le == p || (le == p + 1 && *p == '\r')) { p = nl ? nl + 1 : e; continue; }       // blank line
    if (*p != '>') { fprintf(stderr, "bad FASTA: expected '>'\n"); exit(1); }
    std::string label(p + 1, le);
    if (label.find('\r') != std::string::npos) label.erase(std::remove(label.begin(), label.end(), '\r'), label.end());
    p = nl ? nl + 1 : e;
    const size_t start = out.letters.size();
    while (p < e && *p != '>') {
      nl = (const char *)memchr(p, '\n', (size_t)(e - p));
      le = nl ? nl : e;
      bool pure = true;
      for (const char *c = p; c < le; ++c) pure &= (unsigned char)(((unsigned char)*c | 0x20) - 'a') < 26;
      if (pure) out.letters.append(p, le);
      else for (const char *c = p; c < le; ++c) if ((*c >= 'A' && *c <= 'Z') || (*c >= 'a' && *c <= 'z')) out.letters.push_back(*c);
      p = nl ? nl + 1 : e;
    }
    if (out.letters.size() == start) continue;                  // empty sequence: skipped
    out.labels.push_back(std::move(label)); out.offs.push_back(out.letters.size());
  }
}
static bool parse_fasta_parallel(const char *path, SeqSet &out)
{
  FILE *f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(1); }
  fseek(f, 0, SEEK_END);
  const long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (sz < (8 << 20)) { fclose(f); return false; }
  std::vector<char> buf((size_t)sz);
  if (fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) { fprintf(stderr, "read error on %s\n", path); exit(1); }
  fclose(f);
  if (buf[0] != '>') return false;                                  // FASTQ (or garbage): the serial reader decides
  const unsigned nt = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
  std::vector<const char *> cut(nt + 1);
  const char *b = buf.data(), *e = b + sz;
  cut[0] = b; cut[nt] = e;
  for (unsigned t = 1; t < nt; ++t) {
    const char *p = b + (size_t)sz * t / nt;
    while (p < e && !(p[0] == '\n' && p + 1 < e && p[1] == '>')) ++p;   // the next record start
    cut[t] = p < e ? p + 1 : e;
  }
  std::vector<SeqSet> part(nt);
  std::vector<std::thread> th;
  for (unsigned t = 1; t < nt; ++t) th.emplace_back([&, t] { parse_fasta_piece(cut[t], cut[t + 1], part[t]); });
  parse_fasta_piece(cut[0], cut[1], part[0]);
  for (auto &x : th) x.join();
  size_t nl = 0, ns = 0;
  for (const SeqSet &q : part) { nl += q.letters.size(); ns += q.size(); }
  out.letters.reserve(out.letters.size() + nl); out.labels.reserve(out.labels.size() + ns); out.offs.reserve(out.offs.size() + ns);
  for (SeqSet &q : part) {
    const uint64_t base = out.letters.size();
    out.letters += q.letters;
    for (size_t i = 0; i < q.size(); ++i) { out.labels.push_back(std::move(q.labels[i])); out.offs.push_back(base + q.offs[i + 1]); }
  }
  return true;
}

// Searcher (searcher.h:21-96) as a batch object over one ugs_db
class Searcher {
 public:
  Searcher(const ugs_params &p, const SeqSet &db, int device) : p_(p) {
    if (ugs_db_create(&p_, db.letters.data(), db.offs.data(), (uint32_t)db.size(), device, &db_) != UGS_OK) die("ugs_db_create");
  }
  ~Searcher() { if (b_) ugs_batch_destroy(b_); ugs_db_destroy(db_); }
  const ugs_db *handle() const { return db_; }
  bool pair_keys() const { return p_.pair_mask != 0 || (p_.filter_mask & UGS_F_ABSKEW) != 0; }
  // labels as integer keys (equal labels <=> equal keys) and ;size= annotations (label.cpp:152-161) for the pair filters
  void keys_of(const SeqSet &s, std::vector<uint32_t> &key, std::vector<uint32_t> &size) {
    key.resize(s.size()); size.resize(s.size());
    for (size_t i = 0; i < s.size(); ++i) {
      auto it = label_ids_.find(s.labels[i]);
      if (it == label_ids_.end()) it = label_ids_.emplace(s.labels[i], (uint32_t)label_ids_.size()).first;
      key[i] = it->second;
      const char *z = strstr(s.labels[i].c_str(), ";size=");
      size[i] = z ? (uint32_t)atoi(z + 6) : 0xffffffffu;
    }
  }
  void SetDbKeys(const SeqSet &db) {
    std::vector<uint32_t> k, z;
    keys_of(db, k, z);
    if (ugs_db_set_pair_keys(db_, k.data(), z.data()) != UGS_OK) die("ugs_db_set_pair_keys");
  }
  // upload + search + sync, results left on the device (multi-GPU: they are gathered with ugs_gather_results); q may be empty
  ugs_batch *SearchOnDevice(const SeqSet &q) {
    const uint32_t nq = (uint32_t)q.size();
    if (!b_ || nq > bq_ || q.letters.size() > bl_) {
      if (b_) ugs_batch_destroy(b_);
      bq_ = std::max<uint32_t>(std::max<uint32_t>(nq, 1), bq_); bl_ = std::max<uint64_t>(std::max<uint64_t>(q.letters.size(), 1), bl_);
      if (ugs_batch_create(db_, bq_, bl_, &b_) != UGS_OK) die("ugs_batch_create");
    }
    int rc = ugs_batch_upload(b_, q.letters.data(), q.offs.data(), nq);
    if (rc == UGS_OK) rc = ugs_batch_search(b_);
    if (rc == UGS_OK) rc = ugs_batch_sync(b_);
    if (rc != UGS_OK) die("search");
    return b_;
  }
  void Search(const SeqSet &q, std::vector<ugs_hit> &hits, std::vector<uint32_t> &nhits, std::vector<uint32_t> &pool) {
    const uint32_t nq = (uint32_t)q.size();
    hits.resize((size_t)nq * (p_.max_accepts ? p_.max_accepts : 64) * (p_.strand_both ? 2 : 1) * (p_.local ? p_.max_hsps : 1) + 1);
    nhits.assign(nq + 1, 0);
    pool.resize(24 * (size_t)nq + 4096);                      // grown to the library's demand when a batch needs more (UGS_E_CAPACITY)
    uint64_t used = 0;
    // UGS_E_CAPACITY: the run pool's demand comes back in `used`; otherwise the hit array was too small (unlimited accepts: any number per query)
    auto fetch = [&](ugs_batch *bb) -> int {
      int rc = UGS_E_CAPACITY;
      for (int t = 0; t < 12 && rc == UGS_E_CAPACITY; ++t) {
        rc = ugs_batch_fetch(bb, hits.data(), hits.size(), nhits.data(), pool.data(), pool.size(), &used);
        if (rc == UGS_E_CAPACITY) { if (used > pool.size()) pool.resize(used + 1024); else hits.resize(hits.size() * 4 + 1024); }
      }
      return rc;
    };
    if (pair_keys()) {                                       // staged calls: the one-shot entry point has no room for per-query keys
      std::vector<uint32_t> k, z;
      keys_of(q, k, z);
      ugs_batch *b = nullptr;
      if (ugs_batch_create(db_, nq, q.letters.size(), &b) != UGS_OK) die("ugs_batch_create");
      int rc = ugs_batch_upload(b, q.letters.data(), q.offs.data(), nq);
      if (rc == UGS_OK) rc = ugs_batch_set_pair_keys(b, k.data(), z.data());
      if (rc == UGS_OK) rc = ugs_batch_search(b);
      if (rc == UGS_OK) rc = ugs_batch_sync(b);
      if (rc == UGS_OK) rc = fetch(b);
      ugs_batch_destroy(b);
      if (rc != UGS_OK) die("search with pair filters");
      return;
    }
    // one batch object serves every batch of the run (its device buffers are sized once, for the largest batch seen)
    if (!b_ || nq > bq_ || q.letters.size() > bl_) {
      if (b_) ugs_batch_destroy(b_);
      bq_ = std::max<uint32_t>(nq, bq_); bl_ = std::max<uint64_t>(q.letters.size(), bl_);
      if (ugs_batch_create(db_, bq_, bl_, &b_) != UGS_OK) die("ugs_batch_create");
    }
    int rc = ugs_batch_upload(b_, q.letters.data(), q.offs.data(), nq);
    if (rc == UGS_OK) rc = ugs_batch_search(b_);
    if (rc == UGS_OK) rc = ugs_batch_sync(b_);
    if (rc == UGS_OK) rc = fetch(b_);
    if (rc != UGS_OK) die("search");
  }
 private:
  [[noreturn]] static void die(const char *what) { fprintf(stderr, "%s: %s\n", what, ugs_last_error()); exit(1); }
  ugs_params p_; ugs_db *db_ = nullptr;
  ugs_batch *b_ = nullptr; uint32_t bq_ = 0; uint64_t bl_ = 0;
  std::unordered_map<std::string, uint32_t> label_ids_;
};

// Stages of the driver run as threads (FASTA parsing | GPU search | text formatting) joined by small queues, so that a run
// costs about its slowest stage instead of the sum (the reference overlaps them with its worker threads, search.cpp:121-128)
template <class T> class Channel {
 public:
  explicit Channel(size_t cap) : cap_(cap) {}
  void push(T v) { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return q_.size() < cap_; }); q_.push(std::move(v)); cv_.notify_all(); }
  bool pop(T &v) {          // false: closed and drained
    std::unique_lock<std::mutex> l(m_);
    cv_.wait(l, [&] { return !q_.empty() || closed_; });
    if (q_.empty()) return false;
    v = std::move(q_.front()); q_.pop(); cv_.notify_all();
    return true;
  }
  void close() { std::lock_guard<std::mutex> l(m_); closed_ = true; cv_.notify_all(); }
 private:
  std::mutex m_; std::condition_variable cv_; std::queue<T> q_; size_t cap_; bool closed_ = false;
};
struct SearchResult { std::unique_ptr<SeqSet> q; std::vector<ugs_hit> hits; std::vector<uint32_t> nhits, pool; };

// the optional sinks of one search (OutputSink::OpenOutputFiles outputsink.cpp:60-130, DBHitSink dbhitsink.cpp)
struct Outputs {
  FILE *b6 = nullptr, *uc = nullptr, *user = nullptr, *matched = nullptr, *notmatched = nullptr, *aln = nullptr, *pairs = nullptr, *qseg = nullptr, *tseg = nullptr, *trim = nullptr, *matchedfq = nullptr, *notmatchedfq = nullptr;
  std::string userfields;
  bool output_no_hits = false, top_hit_only = false, top_hits_only = false;
  uint32_t maxhits = 0;
  ugs_otutab *otutab = nullptr; FILE *map = nullptr;   // OTUTableSink
  ugs_closedref *closedref = nullptr; FILE *tabbed = nullptr;   // ClosedRefSink
  std::vector<uint32_t> db_hit_counts;      // DBHitSink::m_HitCounts
  const char *db_masked = nullptr;          // DB letters as the reference holds them (masked)
};

// SeqToFastq seqdb.cpp:14-29 (-matchedfq / -notmatchedfq; "Cannot convert FASTA to FASTQ" without qualities)
static void write_fastq(FILE *f, const SeqSet &q, uint32_t qi)
{
  if (q.quals.empty()) { fprintf(stderr, "Cannot convert FASTA to FASTQ\n"); exit(1); }
  const size_t a = q.offs[qi], n = q.offs[qi + 1] - a;
  fprintf(f, "@%s\n%.*s\n+\n%.*s\n", q.labels[qi].c_str(), (int)n, q.letters.data() + a, (int)n, q.quals.data() + a);
}

// OutputSink::OnQueryDone (outputsink.cpp:358-384): hits of one query, or the no-hit records
static void output_query(Outputs &O, const ugs_params &p, const SeqSet &q, const SeqSet &db, uint32_t qi,
                         const ugs_hit *h, uint32_t n_all, const uint32_t *pool)
{
  static std::vector<char> line(1 << 20);
  const uint32_t ql = (uint32_t)(q.offs[qi + 1] - q.offs[qi]);
  const char *qs = q.letters.data() + q.offs[qi], *qlab = q.labels[qi].c_str();
  auto put_to = [&](FILE *f, int len) {
    if (len < 0) { fprintf(stderr, "%s\n", ugs_last_error()); exit(1); }
    if ((size_t)len >= line.size()) { fprintf(stderr, "output line too long\n"); exit(1); }
    if (f && len > 0) fputs(line.data(), f);
  };
  auto put = [&](FILE *f, int len) {
    if (len < 0) { fprintf(stderr, "%s\n", ugs_last_error()); exit(1); }
    if ((size_t)len >= line.size()) { fprintf(stderr, "output line too long\n"); exit(1); }
    fputs(line.data(), f);
  };
  uint32_t first = 0;
  const uint32_t n = ugs_hits_to_report(h, n_all, O.maxhits, O.top_hit_only, O.top_hits_only, &first);   // HitMgr::GetHitCount
  if (O.otutab) {                                                     // OTUTableSink::OnQueryDone otutabsink.cpp:31-58
    uint32_t top = 0;
    if (n) ugs_hits_to_report(h, n_all, 0, 1, 0, &top);               // HitMgr::GetTopHit
    put_to(O.map, ugs_otutab_add(O.otutab, qlab, n ? db.labels[h[top].target].c_str() : nullptr, line.data(), (int)line.size()));
  }
  if (O.closedref) {                                                  // ClosedRefSink::OnQueryDone closedrefsink.cpp:33-118 (all raw hits)
    std::vector<const char *> tls(n_all);
    for (uint32_t j = 0; j < n_all; ++j) tls[j] = db.labels[h[j].target].c_str();
    put_to(O.tabbed, ugs_closedref_add(O.closedref, qlab, h, n_all, tls.data(), line.data(), (int)line.size()));
  }
  h += first;
  if (n == 0) {                                                       // OutputMatchedFalse outputsink.cpp:392-403
    if (O.uc) put(O.uc, ugs_format_uc_nohit(ql, qlab, line.data(), (int)line.size()));
    if (O.output_no_hits) {
      if (O.b6) put(O.b6, ugs_format_blast6_nohit(qlab, line.data(), (int)line.size()));
      if (O.user) put(O.user, ugs_format_userout(nullptr, nullptr, p.is_nucleo, O.userfields.c_str(), qlab, nullptr, qs, ql, nullptr, 0, line.data(), (int)line.size()));
    }
    if (O.notmatched) put(O.notmatched, ugs_format_fasta(qlab, qs, ql, line.data(), (int)line.size()));
    if (O.notmatchedfq) write_fastq(O.notmatchedfq, q, qi);
    return;
  }
  if (O.aln) {                                                        // OutputReport outputsink.cpp:338-356
    std::vector<const char *> tls(n);
    for (uint32_t j = 0; j < n; ++j) tls[j] = db.labels[h[j].target].c_str();
    put(O.aln, p.local ? ugs_format_alnout_header_local(&p, h, n, qlab, tls.data(), line.data(), (int)line.size())
                       : ugs_format_alnout_header(h, n, qlab, tls.data(), line.data(), (int)line.size()));
  }
  for (uint32_t j = 0; j < n; ++j) {
    const uint32_t t = h[j].target;
    const char *tl = db.labels[t].c_str();
    if (O.trim) put(O.trim, ugs_format_trimout(&h[j], pool, qlab, qs, ql, line.data(), (int)line.size()));
    if (O.pairs || O.qseg || O.tseg) {
      const char *tsq = O.db_masked + db.offs[t]; const uint32_t tlen = (uint32_t)(db.offs[t + 1] - db.offs[t]);
      if (O.pairs) put(O.pairs, ugs_format_fastapairs(&h[j], pool, qlab, tl, qs, ql, tsq, tlen, line.data(), (int)line.size()));
      if (O.qseg) put(O.qseg, ugs_format_segout(&h[j], pool, 0, qlab, tl, qs, ql, tsq, tlen, line.data(), (int)line.size()));
      if (O.tseg) put(O.tseg, ugs_format_segout(&h[j], pool, 1, qlab, tl, qs, ql, tsq, tlen, line.data(), (int)line.size()));
    }
    if (O.aln) put(O.aln, p.local ? ugs_format_alnout_hit_local(&p, &h[j], pool, qlab, tl, qs, ql, O.db_masked + db.offs[t],
                                                                (uint32_t)(db.offs[t + 1] - db.offs[t]), line.data(), (int)line.size())
                                  : ugs_format_alnout_hit(&h[j], pool, p.is_nucleo, qlab, tl, qs, ql, O.db_masked + db.offs[t],
                                                          (uint32_t)(db.offs[t + 1] - db.offs[t]), line.data(), (int)line.size()));
    if (O.b6) put(O.b6, p.local ? ugs_format_blast6_local(&p, &h[j], qlab, tl, line.data(), (int)line.size())
                                : ugs_format_blast6(&h[j], qlab, tl, line.data(), (int)line.size()));
    if (O.uc) put(O.uc, ugs_format_uc_hit(&h[j], pool, p.is_nucleo, qlab, tl, line.data(), (int)line.size()));
    if (O.user) put(O.user, p.local ? ugs_format_userout_local(&p, &h[j], pool, O.userfields.c_str(), qlab, tl, qs, ql, O.db_masked + db.offs[t],
                                                               (uint32_t)(db.offs[t + 1] - db.offs[t]), line.data(), (int)line.size())
                                    : ugs_format_userout(&h[j], pool, p.is_nucleo, O.userfields.c_str(), qlab, tl, qs, ql,
                                                         O.db_masked + db.offs[t], (uint32_t)(db.offs[t + 1] - db.offs[t]), line.data(), (int)line.size()));
    if (!O.db_hit_counts.empty() && !(O.otutab && j > 0)) ++O.db_hit_counts[t];   // DBHitSink::OnQueryDone dbhitsink.cpp:117-140 (otutab: first hit only, :137)
  }
  if (O.matched) put(O.matched, ugs_format_fasta(qlab, qs, ql, line.data(), (int)line.size()));
  if (O.matchedfq) write_fastq(O.matchedfq, q, qi);
}

// LoadUDB (loaddb.cpp:100-125): a .udb is recognised by its magic; its letters are used as stored (already masked)
static bool is_udb_file(const char *path)
{
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  unsigned char m[4] = {0, 0, 0, 0};
  const size_t n = fread(m, 1, 4, f);
  fclose(f);
  return n == 4 && m[0] == 'F' && m[1] == 'B' && m[2] == 'D' && m[3] == 'U';
}

static bool load_udb(const char *path, SeqSet &db, bool &nucleo, uint32_t &word_len)
{
  ugs_udb_info info;
  if (ugs_udb_stat(path, &info) != UGS_OK) { fprintf(stderr, "%s\n", ugs_last_error()); return false; }
  db.letters.resize(info.nletters);
  db.offs.assign(info.nseq + 1, 0);
  std::string labels(info.label_bytes, '\0');
  if (ugs_udb_read(path, &db.letters[0], db.offs.data(), &labels[0], nullptr, nullptr) != UGS_OK) { fprintf(stderr, "%s\n", ugs_last_error()); return false; }
  db.labels.clear();
  for (size_t o = 0; o < labels.size(); o += strlen(labels.c_str() + o) + 1) db.labels.emplace_back(labels.c_str() + o);
  nucleo = info.is_nucleo != 0; word_len = info.word_len;
  return db.labels.size() == info.nseq;
}

static bool guess_nucleo(const SeqSet &db)     // SeqDB::GetIsNucleo samples 100 letters (seqdb.cpp:268-320); here: the first 1000
{
  size_t n = 0, nt = 0;
  for (char c : db.letters) {
    if (n >= 1000) break;
    ++n;
    switch (c | 0x20) { case 'a': case 'c': case 'g': case 't': case 'u': case 'n': ++nt; }
  }
  return n > 0 && nt * 10 >= n * 9;
}

#include <chrono>
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
  const double t_start = now_s();
  const bool prof = getenv("UGS_CLI_PROFILE") != nullptr;
  std::string qpath, dbpath, b6path, ucpath, strand, makeudb, outpath, userpath, matchedpath, notmatchedpath, dbmatchedpath, dbnotmatchedpath;
  std::string tabbedout, trimpath, matchedfqpath, notmatchedfqpath; bool closedref_cmd = false;
  std::string biomout;
  std::string clusterfast, centroidspath, sortname; bool sizein = false, sizeout = false; long minsize = 0;
  std::string otutabout, mapout, alnpath, pairspath, qsegpath, tsegpath; bool otutab_cmd = false; long stepwords = -1;
  ugs_params filt; memset(&filt, 0, sizeof filt);                     // only the filter fields are used
  Outputs O;
  bool hardmask = false;
  bool local_cmd = false; double evalue = -1; double xdrop_u = -1, xdrop_g = -1, ka_dbsize = -1; long maxhsps = -1, hspw = -1;
  int ngpus = 1;                                                     // -gpus N: devices device .. device+N-1, one host thread each
  double id = -1; int maxacc = -1, maxrej = -1, device = 0; long big = -1; size_t batch = 1u << 18; int dbtype = -1;
  long wordlength = -1, bump = -1, minhsp = -1, band = -1; double xdrop_nw = -1, match = 0, mismatch = 0; bool match_set = false, mismatch_set = false;   // index / aligner options
  std::string dbmask;                                               // -dbmask none | user | fastnucleo | fastamino | default (makeudb.cpp:11-25)
  double lopen = -1, lext = -1;                                      // usearch_local gap penalties (positive, as the reference takes them)
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto val = [&]() -> const char * { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(1); } return argv[++i]; };
    if (a == "-makeudb_usearch") makeudb = val(); else if (a == "-output") outpath = val();
    else if (a == "-cluster_fast") clusterfast = val(); else if (a == "-centroids") centroidspath = val();
    else if (a == "-sort") sortname = val(); else if (a == "-sizein") sizein = true; else if (a == "-sizeout") sizeout = true;
    else if (a == "-minsize") minsize = atol(val());
    else if (a == "-otutab") { qpath = val(); otutab_cmd = true; } else if (a == "-otus" || a == "-zotus") dbpath = val();
    else if (a == "-closed_ref") { qpath = val(); closedref_cmd = true; } else if (a == "-tabbedout") tabbedout = val();
    else if (a == "-biomout") biomout = val();
    else if (a == "-otutabout") otutabout = val(); else if (a == "-mapout") mapout = val(); else if (a == "-stepwords") stepwords = atol(val());
    else if (a == "-usearch_local") { qpath = val(); local_cmd = true; } else if (a == "-evalue") evalue = atof(val());
    else if (a == "-xdrop_u") xdrop_u = atof(val()); else if (a == "-xdrop_g") xdrop_g = atof(val()); else if (a == "-ka_dbsize") ka_dbsize = atof(val());
    else if (a == "-maxhsps") maxhsps = atol(val()); else if (a == "-hspw") hspw = atol(val());
    else if (a == "-wordlength") wordlength = atol(val()); else if (a == "-bump") bump = atol(val()); else if (a == "-minhsp") minhsp = atol(val());
    else if (a == "-band") band = atol(val()); else if (a == "-xdrop_nw") xdrop_nw = atof(val());
    else if (a == "-lopen") lopen = atof(val()); else if (a == "-lext") lext = atof(val());
    else if (a == "-dbmask") dbmask = val();
    else if (a == "-match") { match = atof(val()); match_set = true; } else if (a == "-mismatch") { mismatch = atof(val()); mismatch_set = true; }
    else if (a == "-usearch_global") qpath = val(); else if (a == "-db") dbpath = val(); else if (a == "-id") id = atof(val());
    else if (a == "-strand") strand = val(); else if (a == "-blast6out") b6path = val(); else if (a == "-uc") ucpath = val();
    else if (a == "-maxaccepts") maxacc = atoi(val()); else if (a == "-maxrejects") maxrej = atoi(val());
    else if (a == "-big") big = atol(val()); else if (a == "-device") device = atoi(val()); else if (a == "-gpus") ngpus = atoi(val()); else if (a == "-batch") batch = (size_t)atol(val());
    else if (a == "-maxid") { filt.maxid = (float)atof(val()); filt.filter_mask |= UGS_F_MAXID; }
    else if (a == "-mincols") { filt.mincols = (uint32_t)atol(val()); filt.filter_mask |= UGS_F_MINCOLS; }
    else if (a == "-maxgaps") { filt.maxgaps = (uint32_t)atol(val()); filt.filter_mask |= UGS_F_MAXGAPS; }
    else if (a == "-query_cov") { filt.query_cov = (float)atof(val()); filt.filter_mask |= UGS_F_QUERY_COV; }
    else if (a == "-max_query_cov") { filt.max_query_cov = (float)atof(val()); filt.filter_mask |= UGS_F_MAX_QUERY_COV; }
    else if (a == "-target_cov") { filt.target_cov = (float)atof(val()); filt.filter_mask |= UGS_F_TARGET_COV; }
    else if (a == "-max_target_cov") { filt.max_target_cov = (float)atof(val()); filt.filter_mask |= UGS_F_MAX_TARGET_COV; }
    else if (a == "-maxdiffs") { filt.maxdiffs = (uint32_t)atol(val()); filt.filter_mask |= UGS_F_MAXDIFFS; }
    else if (a == "-mindiffs") { filt.mindiffs = (uint32_t)atol(val()); filt.filter_mask |= UGS_F_MINDIFFS; }
    else if (a == "-hardmask") hardmask = true;
    else if (a == "-termid") { filt.termid = (float)atof(val()); filt.align_flags |= UGS_A_TERMID; }
    else if (a == "-termidd") { filt.termidd = (float)atof(val()); filt.align_flags |= UGS_A_TERMIDD; }
    else if (a == "-fulldp") filt.align_flags |= UGS_A_FULLDP; else if (a == "-gaforce") filt.align_flags |= UGS_A_GAFORCE;
    else if (a == "-self") filt.pair_mask |= UGS_P_SELF; else if (a == "-notself") filt.pair_mask |= UGS_P_NOTSELF;
    else if (a == "-selfid") filt.pair_mask |= UGS_P_SELFID;
    else if (a == "-min_sizeratio") { filt.min_sizeratio = (float)atof(val()); filt.pair_mask |= UGS_P_MIN_SIZERATIO; }
    else if (a == "-minqt") { filt.minqt = (float)atof(val()); filt.pair_mask |= UGS_P_MINQT; }
    else if (a == "-maxqt") { filt.maxqt = (float)atof(val()); filt.pair_mask |= UGS_P_MAXQT; }
    else if (a == "-minsl") { filt.minsl = (float)atof(val()); filt.pair_mask |= UGS_P_MINSL; }
    else if (a == "-maxsl") { filt.maxsl = (float)atof(val()); filt.pair_mask |= UGS_P_MAXSL; }
    else if (a == "-abskew") { filt.abskew = (float)atof(val()); filt.filter_mask |= UGS_F_ABSKEW; }
    else if (a == "-alnout") alnpath = val(); else if (a == "-fastapairs") pairspath = val();
    else if (a == "-trimout") trimpath = val();
    else if (a == "-qsegout") qsegpath = val(); else if (a == "-tsegout") tsegpath = val();
    else if (a == "-userout") userpath = val(); else if (a == "-userfields") O.userfields = val();
    else if (a == "-matchedfq") matchedfqpath = val(); else if (a == "-notmatchedfq") notmatchedfqpath = val();
    else if (a == "-matched") matchedpath = val(); else if (a == "-notmatched") notmatchedpath = val();
    else if (a == "-dbmatched") dbmatchedpath = val(); else if (a == "-dbnotmatched") dbnotmatchedpath = val();
    else if (a == "-output_no_hits") O.output_no_hits = true; else if (a == "-top_hit_only") O.top_hit_only = true;
    else if (a == "-top_hits_only") O.top_hits_only = true; else if (a == "-maxhits") O.maxhits = (uint32_t)atol(val());
    else if (a == "-dbtype") { std::string v = val(); dbtype = (v == "nt"); }
    else if (a == "-threads" || a == "-quiet") { if (a == "-threads") val(); }   // accepted, meaningless here
    else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 1; }
  }
  if (!clusterfast.empty()) {                               // cmd_cluster_fast clusterfast.cpp:81-138 (-sort unset: input order)
    if (id < 0) { fprintf(stderr, "Must specify -id\n"); return 1; }                        // makeclustersearcher.cpp:30-31
    SeqSet in;
    { FastaReader r(clusterfast.c_str()); while (r.read(in, 1u << 20)) {} }
    if (in.size() == 0) { fprintf(stderr, "No sequences in input file\n"); return 1; }      // clusterfast.cpp:91-92
    const bool cl_nucleo = dbtype >= 0 ? dbtype != 0 : guess_nucleo(in);                     // (protein input: UCLUST's other everyday use)
    ugs_params p;
    ugs_params_init(&p, cl_nucleo ? 1 : 0, id);
    ugs_params_set_cluster(&p);
    if (!strand.empty()) {                                                                   // StrandOptToRevComp(false, false) clusterfast.cpp:18-36
      if (strand == "both") p.strand_both = 1; else if (strand != "plus") { fprintf(stderr, "Invalid -strand\n"); return 1; }
    }
    if (maxacc >= 0) p.max_accepts = maxacc;
    if (maxrej >= 0) p.max_rejects = maxrej;
    if (big >= 0) p.big = (uint32_t)big;
    int sort_mode = UGS_SORT_NONE;                                                           // GetSeqOrder clusterfast.cpp:37-66
    if (sortname == "length") sort_mode = UGS_SORT_LENGTH; else if (sortname == "size") sort_mode = UGS_SORT_SIZE;
    else if (sortname == "other") { fprintf(stderr, "-cluster_fast does not support -sort other, use -cluster_smallmem\n"); return 1; }
    else if (!sortname.empty() && sortname != "user") { fprintf(stderr, "Invalid sort name %s\n", sortname.c_str()); return 1; }
    std::vector<uint32_t> size_in(in.size());
    for (size_t i = 0; i < in.size(); ++i) size_in[i] = ugs_label_size(in.labels[i].c_str());
    ugs_cluster *c = nullptr;
    if (ugs_cluster_fast_sorted(&p, in.letters.data(), in.offs.data(), (uint32_t)in.size(), sort_mode, size_in.data(), sizein, device, &c) != UGS_OK) { fprintf(stderr, "%s\n", ugs_last_error()); return 1; }
    std::string labels;
    for (const std::string &l : in.labels) { labels += l; labels.push_back('\0'); }
    int rc = 0;
    if (!ucpath.empty() && ugs_cluster_write_uc(c, labels.data(), ucpath.c_str()) != UGS_OK) { fprintf(stderr, "%s\n", ugs_last_error()); rc = 1; }
    if (!centroidspath.empty() && ugs_cluster_write_centroids_sized(c, labels.data(), centroidspath.c_str(), (sizein ? UGS_SIZEIN : 0) | (sizeout ? UGS_SIZEOUT : 0), (uint32_t)std::max(0l, minsize)) != UGS_OK) { fprintf(stderr, "%s\n", ugs_last_error()); rc = 1; }
    uint32_t nu = 0, nc = 0;
    ugs_cluster_counts(c, &nu, &nc, nullptr, nullptr);
    fprintf(stderr, "%zu seqs, %u uniques, %u clusters\n", in.size(), nu, nc);
    ugs_cluster_destroy(c);
    return rc;
  }
  if (!makeudb.empty()) {                                   // cmd_makeudb_usearch makeudb.cpp:27-66
    if (outpath.empty()) { fprintf(stderr, "-makeudb_usearch needs -output\n"); return 1; }
    SeqSet db;
    { FastaReader r(makeudb.c_str()); while (r.read(db, 1u << 20)) {} }
    if (db.size() == 0) { fprintf(stderr, "Empty database\n"); return 1; }
    const bool nucleo = dbtype >= 0 ? dbtype != 0 : guess_nucleo(db);
    ugs_params p;
    ugs_params_init(&p, nucleo, 0.5);
    if (wordlength > 0) p.word_len = (int32_t)wordlength;
    Searcher s(p, db, device);
    std::string labels;
    for (const std::string &l : db.labels) { labels += l; labels.push_back('\0'); }
    if (ugs_udb_write(outpath.c_str(), s.handle(), labels.data(), labels.size()) != UGS_OK) { fprintf(stderr, "%s\n", ugs_last_error()); return 1; }
    return 0;
  }
  if (qpath.empty() || dbpath.empty()) { fprintf(stderr, "usage: ugs_cli -usearch_global q.fa -db db.fa -id 0.97 -strand plus -blast6out o.b6 -uc o.uc\n"); return 1; }
  // stage 1: the query file is parsed by its own thread, batch by batch, from the start (while the DB is read and indexed)
  // (the thread owns what it touches - path, batch size, the channel - so that an early error exit of main is harmless)
  auto parsed_ptr = std::make_shared<Channel<std::unique_ptr<SeqSet>>>(3);
  Channel<std::unique_ptr<SeqSet>> &parsed = *parsed_ptr;
  std::thread([parsed_ptr, qpath, batch] {
    FastaReader qr(qpath.c_str());
    for (;;) {
      std::unique_ptr<SeqSet> q(new SeqSet);
      if (!qr.read(*q, batch)) break;
      parsed_ptr->push(std::move(q));
    }
    parsed_ptr->close();
  }).detach();
  SeqSet db;
  bool from_udb = false, udb_nucleo = true; uint32_t udb_word = 0;
  if (is_udb_file(dbpath.c_str())) {
    if (!load_udb(dbpath.c_str(), db, udb_nucleo, udb_word)) return 1;
    from_udb = true;
  } else if (!parse_fasta_parallel(dbpath.c_str(), db)) { FastaReader r(dbpath.c_str()); while (r.read(db, 1u << 20)) {} }
  if (prof) fprintf(stderr, "[cli] db parsed %.3f s\n", now_s() - t_start);
  const bool nucleo = from_udb ? udb_nucleo : (dbtype >= 0 ? dbtype != 0 : guess_nucleo(db));
  if (otutab_cmd) {                                                   // cmd_otutab searchcmd.cpp:20-40 (oset_*d: only if not given)
    if (id < 0) id = 0.97;
    if (maxacc < 0) maxacc = 3;
    if (maxrej < 0) maxrej = 32;
    if (stepwords < 0) stepwords = 0;
    if (strand.empty()) strand = "both";
  }
  if (closedref_cmd) {                                                // cmd_closed_ref searchcmd.cpp:11-19, terminator.cpp:16-20
    if (id < 0) id = 0.97;
    if (stepwords < 0) stepwords = 0;
    if (maxacc < 0) maxacc = 4;
    if (maxrej < 0) maxrej = 16;
  }
  if (nucleo && strand.empty()) { fprintf(stderr, "-strand plus|both required for a nucleotide db\n"); return 1; }   // search.cpp:23-34
  // no -id: the reference neither dies nor filters by identity - word counting runs with 0.5 (makedbsearcher.cpp:195), Accepter::IsAcceptLo
  // tests the identity only when the option was given (accepter.cpp:35); golden runs hard_noid / hard_noid_s.  cluster_fast does die (above).
  ugs_params p;
  ugs_params_init(&p, nucleo, id < 0 ? 0.5 : id);
  p.id_set = id >= 0;
  if (local_cmd) {                                                    // cmd_usearch_local searchcmd.cpp:42-45; -evalue is required (localaligner.cpp:200)
    if (evalue <= 0) { fprintf(stderr, "Required option not set -evalue\n"); return 1; }
    if (ugs_params_set_local(&p, evalue, id >= 0) != UGS_OK) { fprintf(stderr, "%s\n", ugs_last_error()); return 1; }
    if (xdrop_u >= 0) p.xdrop_u = (float)xdrop_u;
    if (xdrop_g >= 0) p.xdrop_g = (float)xdrop_g;
    if (ka_dbsize > 0) p.ka_dbsize = (float)ka_dbsize;
    if (maxhsps > 0) p.max_hsps = (uint32_t)maxhsps;
    if (lopen >= 0) p.local_open = (float)-lopen;
    if (lext >= 0) p.local_ext = (float)-lext;
    if (!pairspath.empty() || !qsegpath.empty() || !tsegpath.empty() || !trimpath.empty() || otutab_cmd || closedref_cmd) {
      fprintf(stderr, "-usearch_local writes -blast6out, -uc, -userout, -alnout, -matched/-notmatched and -dbmatched/-dbnotmatched only\n"); return 1;
    }
  }
  if (hspw > 0) p.hsp_word_len = (int32_t)hspw;
  if (wordlength > 0) p.word_len = (int32_t)wordlength;
  if (bump >= 0) p.bump_pct = (uint32_t)bump;
  if (minhsp >= 0) p.minhsp = (int32_t)minhsp;
  if (band >= 0) p.band = (int32_t)band;
  if (xdrop_nw >= 0) p.xdrop_nw = (float)xdrop_nw;
  if (match_set) p.match = (float)match;
  if (mismatch_set) p.mismatch = (float)mismatch;
  p.strand_both = nucleo && strand == "both";
  if (maxacc >= 0) p.max_accepts = maxacc;
  if (maxrej >= 0) p.max_rejects = maxrej;
  if (big >= 0) p.big = (uint32_t)big;
  if (stepwords >= 0) p.stepwords = (uint32_t)stepwords;
  p.filter_mask = filt.filter_mask; p.maxid = filt.maxid; p.mincols = filt.mincols; p.maxgaps = filt.maxgaps;
  p.query_cov = filt.query_cov; p.max_query_cov = filt.max_query_cov; p.target_cov = filt.target_cov;
  p.max_target_cov = filt.max_target_cov; p.maxdiffs = filt.maxdiffs; p.mindiffs = filt.mindiffs;
  p.pair_mask = filt.pair_mask; p.min_sizeratio = filt.min_sizeratio; p.minqt = filt.minqt; p.maxqt = filt.maxqt; p.minsl = filt.minsl;
  p.maxsl = filt.maxsl; p.abskew = filt.abskew; p.align_flags = filt.align_flags; p.termid = filt.termid; p.termidd = filt.termidd;
  if (!dbmask.empty()) {
    std::string m = dbmask; for (char &ch : m) ch = (char)tolower((unsigned char)ch);
    if (m == "none") p.dbmask = 0; else if (m == "user") p.dbmask = 2;
    else if (m == "default" || m == (nucleo ? "fastnucleo" : "fastamino")) p.dbmask = 1;
    else { fprintf(stderr, "-dbmask %s is not supported (none, user, default, %s)\n", dbmask.c_str(), nucleo ? "fastnucleo" : "fastamino"); return 1; }
  }
  if (hardmask) p.dbmask = 3;
  if (from_udb) { p.dbmask = 2; p.word_len = (int32_t)udb_word; }   // stored letters are the masked ones (makeudb.cpp:54)
  auto open_out = [](const std::string &path) -> FILE * {
    if (path.empty()) return nullptr;
    FILE *f = fopen(path.c_str(), "w");
    if (!f) { fprintf(stderr, "cannot create %s\n", path.c_str()); exit(1); }
    setvbuf(f, nullptr, _IOFBF, 4u << 20);
    return f;
  };
  if (!userpath.empty()) {                                            // outputsink.cpp:96-103
    if (O.userfields.empty()) { fprintf(stderr, "--userout requires --userfields\n"); return 1; }
    if (ugs_userfields_check(O.userfields.c_str()) != UGS_OK) { fprintf(stderr, "%s\n", ugs_last_error()); return 1; }
  }
  O.b6 = open_out(b6path); O.uc = open_out(ucpath); O.user = open_out(userpath); O.aln = open_out(alnpath); O.pairs = open_out(pairspath); O.trim = open_out(trimpath); O.matchedfq = open_out(matchedfqpath); O.notmatchedfq = open_out(notmatchedfqpath); O.qseg = open_out(qsegpath); O.tseg = open_out(tsegpath);
  O.matched = open_out(matchedpath); O.notmatched = open_out(notmatchedpath);
  if (otutab_cmd) { O.otutab = ugs_otutab_create(); O.map = open_out(mapout); }
  if (closedref_cmd) { O.closedref = ugs_closedref_create(); O.tabbed = open_out(tabbedout); }
  Searcher searcher(p, db, device);
  if (prof) fprintf(stderr, "[cli] db on device %.3f s\n", now_s() - t_start);
  if (searcher.pair_keys()) searcher.SetDbKeys(db);
  std::string masked;
  if (O.user || O.aln || O.pairs || O.qseg || O.tseg || !dbmatchedpath.empty() || !dbnotmatchedpath.empty()) {
    masked.resize(db.letters.size());
    if (ugs_db_masked_letters(searcher.handle(), &masked[0]) != UGS_OK) { fprintf(stderr, "%s\n", ugs_last_error()); return 1; }
    O.db_masked = masked.data();
    O.db_hit_counts.assign(db.size(), 0);
  }
  size_t total = 0, with_hit = 0;
  {
    Channel<SearchResult> results(2);
    std::thread writer([&] {                                         // stage 3: the sinks, in query order
      SearchResult r;
      // the usual outputs (-blast6out / -uc only, global hits) are formatted by several threads, each over a contiguous
      // range of queries into its own buffers, and written out in range order; everything else takes the general loop
      const bool plain = !p.local && !O.user && !O.aln && !O.pairs && !O.qseg && !O.tseg && !O.trim && !O.matched && !O.notmatched &&
                         !O.matchedfq && !O.notmatchedfq && !O.otutab && !O.closedref && O.db_hit_counts.empty() && !O.output_no_hits &&
                         !O.top_hit_only && !O.top_hits_only && O.maxhits == 0;
      while (results.pop(r)) {
        if (plain) {
          const uint32_t nq = (uint32_t)r.q->size();
          const unsigned nt = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
          std::vector<size_t> hoff((size_t)nq + 1, 0);
          for (uint32_t qi = 0; qi < nq; ++qi) hoff[qi + 1] = hoff[qi] + r.nhits[qi];
          std::vector<std::string> sb6(nt), suc(nt);
          std::vector<size_t> nwith(nt, 0);
          auto work = [&](unsigned t) {
            std::vector<char> line(1 << 16);
            const uint32_t lo = (uint32_t)((uint64_t)nq * t / nt), hi = (uint32_t)((uint64_t)nq * (t + 1) / nt);
            for (uint32_t qi = lo; qi < hi; ++qi) {
              const char *qlab = r.q->labels[qi].c_str();
              const uint32_t n = r.nhits[qi];
              if (n == 0) {
                if (O.uc) { const int len = ugs_format_uc_nohit((uint32_t)(r.q->offs[qi + 1] - r.q->offs[qi]), qlab, line.data(), (int)line.size()); suc[t].append(line.data(), (size_t)len); }
                continue;
              }
              ++nwith[t];
              for (uint32_t j = 0; j < n; ++j) {
                const ugs_hit *h = &r.hits[hoff[qi] + j];
                const char *tl = db.labels[h->target].c_str();
                if (O.b6) { const int len = ugs_format_blast6(h, qlab, tl, line.data(), (int)line.size()); if (len < 0 || (size_t)len >= line.size()) { fprintf(stderr, "output line too long\n"); exit(1); } sb6[t].append(line.data(), (size_t)len); }
                if (O.uc) {
                  int len = ugs_format_uc_hit(h, r.pool.data(), p.is_nucleo, qlab, tl, line.data(), (int)line.size());
                  if (len >= (int)line.size()) { line.resize((size_t)len + 16); len = ugs_format_uc_hit(h, r.pool.data(), p.is_nucleo, qlab, tl, line.data(), (int)line.size()); }
                  suc[t].append(line.data(), (size_t)len);
                }
              }
            }
          };
          std::vector<std::thread> th;
          for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
          work(0);
          for (auto &x : th) x.join();
          for (unsigned t = 0; t < nt; ++t) {
            if (O.b6 && !sb6[t].empty()) fwrite(sb6[t].data(), 1, sb6[t].size(), O.b6);
            if (O.uc && !suc[t].empty()) fwrite(suc[t].data(), 1, suc[t].size(), O.uc);
            with_hit += nwith[t];
          }
          total += nq;
          continue;
        }
        size_t k = 0;
        for (uint32_t qi = 0; qi < r.q->size(); ++qi) {
          output_query(O, p, *r.q, db, qi, r.hits.data() + k, r.nhits[qi], r.pool.data());
          k += r.nhits[qi]; with_hit += r.nhits[qi] > 0;
        }
        total += r.q->size();
      }
    });
    if (ngpus > 1 || getenv("UGS_CLI_FORCE_GATHER")) {            // (the variable sends a single-GPU run through the gather path: tests)
      // stage 2 on N GPUs: the index is replicated, every round hands N consecutive query batches to the N devices
      // (one host thread each), and the only exchange is the gather of the round's device-resident hit tables to rank 0 over
      // RCCL (include/ugs_comm.h), which arrive in rank order = query order and go to the sinks as one result
      if (searcher.pair_keys()) { fprintf(stderr, "-gpus: pair filters / -abskew are single-GPU options here\n"); exit(1); }
      std::vector<std::unique_ptr<Searcher>> reps((size_t)ngpus);
      {
        std::vector<std::thread> th;
        for (int g = 1; g < ngpus; ++g) th.emplace_back([&, g] { reps[(size_t)g].reset(new Searcher(p, db, device + g)); });
        for (auto &x : th) x.join();
      }
      std::vector<int> devs((size_t)ngpus);
      for (int g = 0; g < ngpus; ++g) devs[(size_t)g] = device + g;
      std::vector<ugs_comm *> comms((size_t)ngpus, nullptr);
      if (ugs_comm_init_all(ngpus, devs.data(), comms.data()) != UGS_OK) { fprintf(stderr, "ugs_comm_init_all: %s\n", ugs_last_error()); exit(1); }
      const SeqSet empty;
      for (bool more = true; more;) {
        std::vector<std::unique_ptr<SeqSet>> qs((size_t)ngpus);
        int got = 0;
        for (; got < ngpus; ++got) if (!parsed.pop(qs[(size_t)got])) { more = false; break; }
        if (got == 0) break;
        std::vector<uint32_t> base((size_t)ngpus + 1, 0);
        for (int g = 0; g < ngpus; ++g) base[(size_t)g + 1] = base[(size_t)g] + (qs[(size_t)g] ? (uint32_t)qs[(size_t)g]->size() : 0u);
        const uint32_t nq = base[(size_t)ngpus];
        SearchResult r;
        r.hits.resize((size_t)nq * (p.max_accepts ? p.max_accepts : 64) * (p.strand_both ? 2 : 1) * (p.local ? p.max_hsps : 1) + 1);
        r.nhits.assign((size_t)nq + 1, 0);
        r.pool.resize(24 * (size_t)nq + 4096);
        std::vector<int> rcs((size_t)ngpus, UGS_OK);
        auto rank_work = [&](int g) {
          Searcher &S = g == 0 ? searcher : *reps[(size_t)g];
          ugs_batch *b = S.SearchOnDevice(qs[(size_t)g] ? *qs[(size_t)g] : empty);
          uint64_t nh = 0, nqt = 0, used = 0;
          int rc = ugs_gather_results(comms[(size_t)g], b, base[(size_t)g], 0, g == 0 ? r.hits.data() : nullptr, g == 0 ? r.hits.size() : 0,
                                      g == 0 ? r.nhits.data() : nullptr, g == 0 ? r.nhits.size() : 0, g == 0 ? r.pool.data() : nullptr,
                                      g == 0 ? r.pool.size() : 0, &nh, &nqt, &used);
          if (g == 0 && rc == UGS_E_CAPACITY) {               // the exchange is complete: only the host copy is repeated with room
            r.hits.resize(nh + 1); r.nhits.resize(nqt + 1); r.pool.resize(used + 1024);
            rc = ugs_gather_refetch(comms[0], r.hits.data(), r.hits.size(), r.nhits.data(), r.nhits.size(), r.pool.data(), r.pool.size(), &nh, &nqt, &used);
          }
          if (rc != UGS_OK) fprintf(stderr, "gather (rank %d): %s\n", g, ugs_last_error());      // (the message is per thread)
          rcs[(size_t)g] = rc;
        };
        std::vector<std::thread> th;
        for (int g = 1; g < ngpus; ++g) th.emplace_back(rank_work, g);
        rank_work(0);
        for (auto &x : th) x.join();
        for (int g = 0; g < ngpus; ++g) if (rcs[(size_t)g] != UGS_OK) exit(1);
        // the round's queries as one set, in rank order
        r.q = std::move(qs[0]);
        for (int g = 1; g < got; ++g) {
          SeqSet &d = *r.q; const SeqSet &a = *qs[(size_t)g];
          const uint64_t off0 = d.letters.size();
          d.letters += a.letters; d.quals += a.quals;
          for (size_t i = 0; i < a.size(); ++i) { d.labels.push_back(a.labels[i]); d.offs.push_back(off0 + a.offs[i + 1]); }
        }
        results.push(std::move(r));
      }
      for (ugs_comm *c : comms) ugs_comm_destroy(c);
    } else
    for (;;) {                                                        // stage 2: the GPU
      std::unique_ptr<SeqSet> q;
      if (!parsed.pop(q)) break;
      SearchResult r;
      if (prof) fprintf(stderr, "[cli] batch popped %.3f s\n", now_s() - t_start);
      searcher.Search(*q, r.hits, r.nhits, r.pool);
      if (prof) fprintf(stderr, "[cli] batch searched %.3f s\n", now_s() - t_start);
      r.q = std::move(q);
      results.push(std::move(r));
    }
    results.close();
    writer.join();
    if (prof) fprintf(stderr, "[cli] written %.3f s\n", now_s() - t_start);
  }
  for (FILE *f : {O.b6, O.uc, O.user, O.matched, O.notmatched, O.map, O.aln, O.pairs, O.qseg, O.tseg, O.tabbed, O.trim, O.matchedfq, O.notmatchedfq}) if (f) fclose(f);
  if (O.closedref) ugs_closedref_destroy(O.closedref);
  if (O.otutab) {                                                     // OTUTableSink::OnAllDone otutabsink.cpp:60-76
    uint64_t assigned = 0, tot = 0;
    ugs_otutab_totals(O.otutab, &assigned, &tot);
    fprintf(stderr, "%llu / %llu mapped to OTUs (%.1f%%)\n", (unsigned long long)assigned, (unsigned long long)tot, tot ? 100.0 * assigned / tot : 0.0);
    if (!otutabout.empty() && ugs_otutab_write(O.otutab, otutabout.c_str()) != UGS_OK) { fprintf(stderr, "%s\n", ugs_last_error()); return 1; }
    if (!biomout.empty() && ugs_otutab_write_biom(O.otutab, biomout.c_str()) != UGS_OK) { fprintf(stderr, "%s\n", ugs_last_error()); return 1; }
    ugs_otutab_destroy(O.otutab);
  }
  for (int m = 0; m < 2; ++m) {                                       // DBHitSink::ToFASTA dbhitsink.cpp:89-115
    const std::string &path = m ? dbmatchedpath : dbnotmatchedpath;
    if (path.empty()) continue;
    FILE *f = open_out(path);
    std::vector<char> rec;
    for (size_t t = 0; t < db.size(); ++t) {
      if ((O.db_hit_counts[t] > 0) != (m == 1)) continue;
      const uint32_t L = (uint32_t)(db.offs[t + 1] - db.offs[t]);
      rec.resize((size_t)L + L / 80 + db.labels[t].size() + 8);
      ugs_format_fasta(db.labels[t].c_str(), O.db_masked + db.offs[t], L, rec.data(), (int)rec.size());
      fputs(rec.data(), f);
    }
    fclose(f);
  }
  fprintf(stderr, "%zu queries, %zu with hits (%.1f%%)\n", total, with_hit, total ? 100.0 * with_hit / total : 0.0);
  return 0;
}
