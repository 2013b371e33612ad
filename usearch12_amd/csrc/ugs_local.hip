// ugs_local.hip - usearch_local on gfx950: the candidate walk of one query strand per wavefront, every candidate
// through LocalAligner2::AlignMulti.  The only caller of the gapped x-drop kernels in the reference (SURVEY.md 8a
// X1-X3, 8f-4).
//
// What it replaces (reference):
//   Searcher::Align, non-global branch     searcher.cpp:28-50   (IsAccept per AR, one accept/reject per TARGET)
//   LocalAligner2::SetQueryImpl/AlignMulti localaligner2.cpp:62-145, localmulti.cpp:9-118
//   LocalAligner::AlignPos + GetAnchor     localaligner.cpp:11-58,101-222
//   LocalAligner2::KeepAR / OverlapFract   localaligner2.cpp:228-246, hsp.h:74-89
//   XDropAlignMem                          xdropalignmem.cpp:26-244 (device code shared with k_xdrop: ugs_xdrop_dev.h)
//   Accepter::IsAcceptLo (local branches)  accepter.cpp:24-91, arscorer.cpp:122-154
//
// Mapping.  The reference walks the target word by word; every query position holding the same word is a seed that
// is extended without gaps (x-drop 16), gated by a minimum raw score, anchored, extended with gaps (x-drop 32) and
// gated by an e-value; an accepted HSP moves the walk past its end.  The ungapped extension and the anchor of a seed
// depend on nothing but the seed, so they are computed speculatively, one seed per lane, 64 seeds at a time in walk
// order; only the few seeds that survive go through the sequential part (gapped extension by the whole wave, overlap
// test against the HSPs kept so far, walk pointer).  The two Karlin-Altschul gates are monotone in the score, so the
// host turns them into per-query integer thresholds (half-units) with the reference's own arithmetic; the device
// never evaluates a logarithm.  Scores are exact half-unit integers as in k_xdrop.
#include "ugs_dev.h"
#include "ugs_xdrop_dev.h"
#include <algorithm>

// phase clocks: tuning builds only (-DUGS_ALIGN_CLOCKS=1; see ugs_align.hip)
#ifndef UGS_ALIGN_CLOCKS
#define UGS_ALIGN_CLOCKS 0
#endif
#define ACLK() (UGS_ALIGN_CLOCKS ? clock64() : 0ull)

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ugs_set_error("%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return UGS_E_HIP; } } while (0)

namespace {

constexpr uint32_t LOC_MAXARS = 32;      // HSPs kept per (query strand, target) for the overlap test
constexpr uint32_t LOC_BT = 256;         // target positions listed per round

struct LocWave {
  uint8_t *Aq;          // query strand, raw letters (what the x-drop code reads)
  uint8_t *Ax;          // x-drop score class of each query letter
  uint32_t *qs;         // (seed word << 16) | query position, sorted; padded with ~0 to a power of two
  uint8_t *Bx;          // x-drop score class of each target letter
  uint32_t *seeds;      // (target pos << 16) | query pos, walk order
  uint32_t *ars;        // kept HSPs: Loi, Loj, Leni, Lenj
};

__device__ __forceinline__ uint64_t rl64(uint64_t v, int lane)
{
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), lane) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
}

}  // namespace

__global__ __launch_bounds__(256) void k_local(UgsDbView db, UgsBatchView bv, UgsLocalView lv, uint32_t wave_lds)
{
  extern __shared__ __align__(16) unsigned char smem[];
  int8_t *s_sub2 = (int8_t *)smem;                     // 1024: x-drop score table
  uint8_t *s_xcls = smem + 1024;                       // 256 : x-drop letter class
  uint8_t *s_cls = smem + 1280;                        // 256 : identity class (letter | 32 lower-case, 31 non-alpha)
  uint8_t *s_comp = smem + 1536;                       // 256
  uint8_t *s_hl = smem + 1792;                         // 256 : seed letter, wildcards -> 0 (localaligner2.cpp:96-98)
  uint64_t *s_match = (uint64_t *)(smem + 2048);       // 512
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wpb = blockDim.x >> 6;
  const UgsTables *tab = db.tab;
  for (int k = tid; k < 1024; k += blockDim.x) s_sub2[k] = lv.sub2[k];
  for (int k = tid; k < 256; k += blockDim.x) { s_xcls[k] = lv.cls[k]; s_cls[k] = tab->cls[k]; s_comp[k] = tab->comp[k]; s_hl[k] = tab->hsp_letter[k]; }
  for (int k = tid; k < 64; k += blockDim.x) s_match[k] = tab->match[k];
  __syncthreads();

  const uint32_t maxq = (bv.max_qlen + 15u) & ~15u;
  unsigned char *wb = smem + 2560 + (size_t)wave * wave_lds;
  XdWave w;
  w.lane = lane;
  w.M0 = (int *)wb; w.M1 = w.M0 + lv.W; w.D = w.M1 + lv.W; w.Bc = (uint8_t *)(w.D + lv.W);
  w.sub2 = s_sub2; w.cls = s_xcls;
  size_t off = ((size_t)lv.W * 13 + 15) & ~(size_t)15;
  LocWave L;
  L.Aq = wb + off; off += maxq;
  L.Ax = wb + off; off += maxq;
  uint32_t q2 = 64; while (q2 < maxq) q2 <<= 1;
  L.qs = (uint32_t *)(wb + off); off += (size_t)q2 * 4;
  L.Bx = wb + off; off += (db.max_tlen + 15u) & ~15u;
  L.seeds = (uint32_t *)(wb + off); off += (size_t)lv.seed_cap * 4;
  L.ars = (uint32_t *)(wb + off);
  const uint32_t gw = blockIdx.x * wpb + (uint32_t)wave;
  w.tb = lv.tb + (size_t)gw * lv.tb_cap;
  w.rowinfo = lv.rowinfo + (size_t)gw * lv.rows_cap;
  uint32_t *runsF = lv.runbuf + (size_t)gw * 3 * lv.runbuf_cap, *runsB = runsF + lv.runbuf_cap, *mruns = runsB + lv.runbuf_cap;
  XdView xv;                                           // the fields xd_extend reads
  xv.tb_cap = lv.tb_cap; xv.runbuf_cap = lv.runbuf_cap; xv.open2 = lv.open2; xv.ext2 = lv.ext2;
  xv.X = lv.xdrop_g; xv.abs_open = lv.abs_open; xv.abs_ext = lv.abs_ext;

  const uint32_t units = bv.nq * bv.nstrand, K = bv.K;
  const uint32_t max_acc = (uint32_t)db.max_accepts, max_rej = (uint32_t)db.max_rejects;
  const uint32_t SW = lv.seed_w, alpha = (uint32_t)db.alpha;
  unsigned long long *ctr = bv.counters;
  unsigned long long cells = 0, w_tletters = 0, w_pairs = 0, w_hits = 0;
  unsigned long long tl0 = 0, tl1 = 0, tl2 = 0, tl3 = 0, tq;     // phase clocks: listing, ungapped + anchor, gapped, hit statistics

  for (;;) {
    uint32_t unit = 0;
    if (lane == 0) unit = (uint32_t)atomicAdd(&ctr[UGS_CTR_NEXT_UNIT], 1ull);
    unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit);
    // deep walks (ugs_deep.hip, as k_align): a continuation pass takes the parked units from walk_units and walks on through their complete lists
    const bool deep = (db.align_flags & UGS_A_DEEP) != 0;
    bool cont = false; uint32_t widx = 0;
    if (bv.walk_units) { if (unit >= bv.n_walk) break; widx = unit; unit = bv.walk_units[widx]; cont = true; }
    if (unit >= units) break;
    const uint32_t qi = unit / bv.nstrand, strand = unit % bv.nstrand;
    const uint64_t qo = bv.qoffs[qi];
    const uint32_t QL = (uint32_t)(bv.qoffs[qi + 1] - qo);
    uint32_t ncand = bv.cand_n[unit];
    const int2 thr = lv.qthr[qi];                      // (ungapped, gapped) minimum scores in half-units
    uint32_t nhit = 0, nacc = 0, nrej = 0;
    uint32_t page = 0, n_total = 0, xhead = 0xffffffffu, xcur = 0xffffffffu;
    const uint64_t *dkeys = nullptr;
    bool walk_ended = false;
    if (cont) {
      const UgsWalkState st = bv.walk_state[unit];
      nacc = st.nacc; nrej = st.nrej; page = st.nvis; nhit = st.pad0;
      dkeys = bv.deep_keys + bv.deep_off[widx];
      n_total = (uint32_t)(bv.deep_off[widx + 1] - bv.deep_off[widx]);
      ncand = page < n_total ? (n_total - page < 64u ? n_total - page : 64u) : 0u;
    }
    const uint32_t nqw = QL > SW ? QL - SW + 1 : 0;    // localaligner2.cpp:72-73: QL <= W leaves the query without words
    if (ncand) {
      for (uint32_t p = lane; p < QL; p += 64) {
        const uint8_t ch = (strand == 0) ? bv.qseqs[qo + p] : s_comp[bv.qseqs[qo + (QL - 1 - p)]];
        L.Aq[p] = ch; L.Ax[p] = s_xcls[ch];
      }
      wave_sync();
      // the query's seed words sorted by (word, position): LocalAligner2::SetQueryImpl's QueryPosVec, localaligner2.cpp:62-145
      uint32_t n2 = 64; while (n2 < nqw) n2 <<= 1;
      for (uint32_t p = lane; p < n2; p += 64) {
        uint32_t key = 0xffffffffu;
        if (p < nqw) {
          uint32_t wd = 0;
          for (uint32_t i = 0; i < SW; ++i) wd = wd * alpha + s_hl[L.Aq[p + i]];
          key = (wd << 16) | p;
        }
        L.qs[p] = key;
      }
      wave_sync();
      if (nqw > 1)
        for (uint32_t kk = 2; kk <= n2; kk <<= 1)
          for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
            for (uint32_t i = lane; i < n2; i += 64) {
              const uint32_t l = i ^ j;
              if (l > i) {
                const uint32_t x = L.qs[i], y = L.qs[l];
                if ((x > y) == ((i & kk) == 0)) { L.qs[i] = y; L.qs[l] = x; }
              }
            }
            wave_sync();
          }
    }
  next_page:
    uint32_t ct = 0, clen = 0; uint64_t cto = 0;
    if ((uint32_t)lane < ncand) {
      ct = cont ? (uint32_t)dkeys[page + (uint32_t)lane] : bv.cand[(uint64_t)unit * K + lane];
      cto = db.offs[ct];
      clen = (uint32_t)(db.offs[ct + 1] - cto);
    }
    for (uint32_t k = 0; k < ncand; ++k) {
      const uint32_t t = (uint32_t)rl((int)ct, (int)k);
      const uint64_t to = rl64(cto, (int)k);
      const uint32_t TL = (uint32_t)rl((int)clen, (int)k);
      const uint8_t *B = db.seqs + to;
      w_tletters += TL; ++w_pairs;
      for (uint32_t p = lane; p < TL; p += 64) L.Bx[p] = s_xcls[B[p]];
      wave_sync();
      bool any_accept = false;
      uint32_t nars = 0;
      if (TL >= 2 * SW && nqw) {                       // localmulti.cpp:17-20
        const uint32_t TWC = TL - SW + 1;
        uint32_t curT = 0;                             // the reference's TargetPos
        uint32_t fLoi = 0xffffffffu, fLoj = 0, fLen = 0;   // last anchor whose gapped extension was not kept
        while (curT < TWC) {
          // ---- list the seeds of target positions [curT, tend) in walk order: position ascending, then query position
          tq = ACLK();
          uint32_t tend = min(curT + LOC_BT, TWC), nseeds = 0;
          for (uint32_t t0 = curT; t0 < tend; t0 += 64) {
            // lane = target position: its word's slice of the sorted query words is its seeds, already in query order
            const uint32_t tp = t0 + lane;
            uint32_t lb = 0, cnt = 0;
            if (tp < tend) {
              uint32_t tw = 0;
              for (uint32_t i = 0; i < SW; ++i) tw = tw * alpha + s_hl[B[tp + i]];
              uint32_t lo = 0, hi = nqw;
              const uint32_t k0 = tw << 16; const uint64_t k1 = (uint64_t)(tw + 1) << 16;
              while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (L.qs[mid] < k0) lo = mid + 1; else hi = mid; }
              lb = lo; hi = nqw;
              while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)L.qs[mid] < k1) lo = mid + 1; else hi = mid; }
              cnt = lo - lb;
            }
            uint32_t incl = cnt;                         // inclusive prefix sum over the lanes
            for (int o = 1; o < 64; o <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += x; }
            const uint64_t over = __ballot(nseeds + incl > lv.seed_cap);
            const int cut = over ? __ffsll((long long)over) - 1 : 64;      // the round ends before this position
            if (lane < cut) for (uint32_t r = 0; r < cnt; ++r) L.seeds[nseeds + incl - cnt + r] = (tp << 16) | (L.qs[lb + r] & 0xffffu);
            if (over) { nseeds += cut ? (uint32_t)rl((int)incl, cut - 1) : 0u; tend = t0 + (uint32_t)cut; break; }
            nseeds += (uint32_t)rl((int)incl, 63);
          }
          if (tend == curT) { atomicOr(&ctr[UGS_CTR_ERR], (unsigned long long)UGS_ERR_LOCAL); break; }   // one position overflowed the list
          wave_sync();
          tl0 += ACLK() - tq;
          uint32_t cur = curT;
          for (uint32_t g0 = 0; g0 < nseeds && cur < tend; g0 += 64) {
            // ---- one seed per lane: ungapped x-drop both ways (localaligner.cpp:107-160) and the anchor (:11-58)
            tq = ACLK();
            const uint32_t s = g0 + lane;
            const uint32_t sd = s < nseeds ? L.seeds[s] : 0;
            const uint32_t st = sd >> 16, sq = sd & 0xffffu;
            bool pass = false;
            uint32_t aLoi = 0, aLoj = 0, aLen = 0;
            if (s < nseeds && st >= cur) {
              int tot = 0, best = 0; uint32_t len = 0, kk = 0;
              int i = (int)sq, j = (int)st;
              while (i >= 0 && j >= 0) {
                ++kk;
                tot += s_sub2[((int)L.Ax[i] << 5) | L.Bx[j]];
                if (tot > best) { best = tot; len = kk; }
                else if ((float)(best - tot) * 0.5f > lv.xdrop_u) break;
                --i; --j;
              }
              const int left = best; const uint32_t leftlen = len;
              tot = 0; best = 0; len = 0; kk = 0;
              i = (int)sq + 1; j = (int)st + 1;
              while (i < (int)QL && j < (int)TL) {
                ++kk;
                tot += s_sub2[((int)L.Ax[i] << 5) | L.Bx[j]];
                if (tot > best) { best = tot; len = kk; }
                else if ((float)(best - tot) * 0.5f > lv.xdrop_u) break;
                ++i; ++j;
              }
              if (left + best >= thr.x) {
                const uint32_t Loi = sq + 1 - leftlen, Loj = st + 1 - leftlen, SegLen = leftlen + len;
                uint32_t startk = 0xffffffffu, beststart = 0xffffffffu, blen = 0;
                int asc = 0, bsc = 0;
                for (uint32_t x = 0; x < SegLen; ++x) {
                  const int sc = s_sub2[((int)L.Ax[Loi + x] << 5) | L.Bx[Loj + x]];
                  if (sc > 0) {
                    if (startk == 0xffffffffu) { startk = x; asc = sc; } else asc += sc;
                  } else {
                    if (asc > bsc) { bsc = asc; beststart = startk; blen = x - startk; }
                    startk = 0xffffffffu;
                  }
                }
                if (asc > bsc) { bsc = asc; beststart = startk; blen = SegLen - startk; }
                if (bsc > 0) { pass = true; aLoi = Loi + beststart; aLoj = Loj + beststart; aLen = blen; }
              }
            }
            // ---- the survivors, in walk order
            uint64_t pm = __ballot(pass);
            tl1 += ACLK() - tq;
            while (pm) {
              const int l = __ffsll((long long)pm) - 1;
              pm &= pm - 1;
              const uint32_t tt = (uint32_t)rl((int)st, l);
              if (tt < cur) continue;
              const uint32_t AncLoi = (uint32_t)rl((int)aLoi, l), AncLoj = (uint32_t)rl((int)aLoj, l), AncLen = (uint32_t)rl((int)aLen, l);
              if (AncLen <= 1) continue;               // xdropalignmem.cpp:44-49: score 0
              if (AncLoi == fLoi && AncLoj == fLoj && AncLen == fLen) continue;   // same extension, same verdict
              // XDropAlignMem xdropalignmem.cpp:26-214
              tq = ACLK();
              const uint32_t AncHii = AncLoi + AncLen - 1, AncHij = AncLoj + AncLen - 1;
              uint32_t bi = 0, bj = 0, fi = 0, fj = 0, nb = 0, nf = 0;
              bool ovf = false;
              const int bwd = xd_side(xv, w, L.Aq + AncLoi, -1, AncLoi + 1, B + AncLoj, -1, AncLoj + 1,
                                      AncLoi > XD_MAXL || AncLoj > XD_MAXL, bi, bj, runsB, nb, ovf, cells);
              int fwd = 0;
              if (!ovf) fwd = xd_side(xv, w, L.Aq + AncHii, 1, QL - AncHii, B + AncHij, 1, TL - AncHij,
                                      (QL - AncHii) > XD_MAXL || (TL - AncHij) > XD_MAXL, fi, fj, runsF, nf, ovf, cells);
              if (ovf || nb > lv.runbuf_cap || nf > lv.runbuf_cap) { atomicOr(&ctr[UGS_CTR_ERR], (unsigned long long)UGS_ERR_LOCAL); continue; }
              int anc = 0;
              for (uint32_t x = lane; x < AncLen; x += 64) anc += s_sub2[((int)L.Ax[AncLoi + x] << 5) | L.Bx[AncLoj + x]];
              for (int o = 32; o; o >>= 1) anc += __shfl_xor(anc, o);
              const int dupe = s_sub2[((int)L.Ax[AncLoi] << 5) | L.Bx[AncLoj]] + s_sub2[((int)L.Ax[AncHii] << 5) | L.Bx[AncHij]];
              const int score2 = bwd + fwd + anc - dupe;                                   // :176
              tl2 += ACLK() - tq; tq = ACLK();
              const uint32_t Loi = AncLoi + 1 - bi, Loj = AncLoj + 1 - bj;
              const uint32_t Leni = bi + fi + AncLen - 2, Lenj = bj + fj + AncLen - 2;
              bool keep = score2 > 0 && score2 >= thr.y;                                  // localaligner.cpp:193-205
              if (keep) {                                                                 // KeepAR
                bool ov = false;
                if ((uint32_t)lane < nars) {
                  const uint32_t oLoi = L.ars[lane * 4], oLoj = L.ars[lane * 4 + 1], oLeni = L.ars[lane * 4 + 2], oLenj = L.ars[lane * 4 + 3];
                  const uint32_t MaxLoi = max(Loi, oLoi), MaxLoj = max(Loj, oLoj);
                  const uint32_t MinHii = min(Loi + Leni - 1, oLoi + oLeni - 1), MinHij = min(Loj + Lenj - 1, oLoj + oLenj - 1);
                  const uint32_t Ovi = MinHii < MaxLoi ? 0 : MinHii - MaxLoi, Ovj = MinHij < MaxLoj ? 0 : MinHij - MaxLoj;
                  ov = 2ull * (uint64_t)(Ovi * Ovj) > (uint64_t)(Leni * Lenj) && Leni && Lenj;   // OverlapFract > 0.5
                }
                if (__ballot(ov)) keep = false;
              }
              if (!keep) { fLoi = AncLoi; fLoj = AncLoj; fLen = AncLen; continue; }
              if (nars >= LOC_MAXARS) { atomicOr(&ctr[UGS_CTR_ERR], (unsigned long long)UGS_ERR_LOCAL); break; }
              if (lane == 0) { L.ars[nars * 4] = Loi; L.ars[nars * 4 + 1] = Loj; L.ars[nars * 4 + 2] = Leni; L.ars[nars * 4 + 3] = Lenj; }
              ++nars;
              // ---- path = reverse(backward side) + M x (AncLen-2) + forward side, adjacent equal ops merged
              uint32_t nm = 0;
              if (lane == 0) {
                uint32_t cop = 0, clen2 = 0;
                auto put = [&](uint32_t r) {
                  if (clen2 && (r & 3u) == cop) clen2 += r >> 2;
                  else { if (clen2) mruns[nm++] = (clen2 << 2) | cop; cop = r & 3u; clen2 = r >> 2; }
                };
                for (uint32_t x = 0; x < nb; ++x) put(xl32(&runsB[nb - 1 - x]));
                if (AncLen > 2) put((AncLen - 2) << 2);
                for (uint32_t x = 0; x < nf; ++x) put(xl32(&runsF[x]));
                if (clen2) mruns[nm++] = (clen2 << 2) | cop;
              }
              nm = (uint32_t)__builtin_amdgcn_readfirstlane((int)nm);
              wave_sync();
              // ---- AlignResult::FillLo on the run list (arscorer.cpp:201-296)
              uint32_t qpos = Loi, tpos = Loj, ids = 0, mcols = 0, gaps = 0, opens = 0, cols = 0, lastop = 0;
              for (uint32_t r = 0; r < nm; ++r) {
                const uint32_t run = xl32(&mruns[r]), op = run & 3u, len = run >> 2;
                if (op == 0) {
                  for (uint32_t x0 = 0; x0 < len; x0 += 64) {
                    const uint32_t x = x0 + lane;
                    const bool idn = x < len && ((s_match[s_cls[L.Aq[qpos + x]]] >> s_cls[B[tpos + x]]) & 1ull);
                    ids += (uint32_t)__popcll(__ballot(idn));
                  }
                  mcols += len; qpos += len; tpos += len;
                } else {
                  gaps += len;
                  if (lastop == 0) ++opens;
                  if (op == 1) qpos += len; else tpos += len;
                }
                cols += len; lastop = op;
              }
              // ---- Accepter::IsAcceptLo, local branches (the e-value was tested above with the same score)
              bool accept = true;
              if (db.id_set) {
                const double FractId = cols == 0 ? 0.0 : (double)ids / (double)cols;
                if (FractId < db.id_accept) accept = false;
                if ((db.filter_mask & UGS_F_MAXID) && FractId > (double)db.maxid) accept = false;
              }
              if (db.filter_mask) {
                const uint32_t fm = db.filter_mask, diffs = (mcols - ids) + gaps;
                if ((fm & UGS_F_MINCOLS) && cols < db.mincols) accept = false;
                if ((fm & UGS_F_MAXGAPS) && gaps > db.maxgaps) accept = false;
                if (fm & (UGS_F_QUERY_COV | UGS_F_MAX_QUERY_COV)) {
                  const double Cov = (double)Leni / (double)QL;                           // arscorer.cpp:126-130
                  if ((fm & UGS_F_QUERY_COV) && Cov < (double)db.query_cov) accept = false;
                  if ((fm & UGS_F_MAX_QUERY_COV) && Cov > (double)db.max_query_cov) accept = false;
                }
                if (fm & (UGS_F_TARGET_COV | UGS_F_MAX_TARGET_COV)) {
                  const double Cov = (double)Lenj / (double)TL;                           // arscorer.cpp:143-147
                  if ((fm & UGS_F_TARGET_COV) && Cov < (double)db.target_cov) accept = false;
                  if ((fm & UGS_F_MAX_TARGET_COV) && Cov > (double)db.max_target_cov) accept = false;
                }
                if ((fm & UGS_F_MAXDIFFS) && diffs > db.maxdiffs) accept = false;
                if ((fm & UGS_F_MINDIFFS) && diffs < db.mindiffs) accept = false;
              }
              if (accept) {
                any_accept = true;
                bool hslot = true;
                if (nhit >= lv.hit_slots) {
                  if (!deep) { atomicOr(&ctr[UGS_CTR_ERR], (unsigned long long)UGS_ERR_LOCAL_HITS); hslot = false; }
                  else {
                    // beyond the unit's slots (deep walks): blocks of UGS_XBLOCK hits chained per unit, as k_align's
                    if ((nhit - lv.hit_slots) % UGS_XBLOCK == 0u) {
                      uint32_t nb = 0;
                      if (lane == 0) nb = (uint32_t)atomicAdd(bv.xblocks_used, 1ull);
                      nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
                      if (nb < bv.xblocks_cap) {
                        if (lane == 0) { bv.xnext[nb] = 0xffffffffu; if (xcur != 0xffffffffu) bv.xnext[xcur] = nb; }
                        if (xcur == 0xffffffffu) xhead = nb;
                        xcur = nb;
                      } else { if (lane == 0) atomicOr(&ctr[UGS_CTR_ERR], (unsigned long long)UGS_ERR_XHITS); xcur = 0xfffffffeu; }
                    }
                    hslot = xcur < 0xfffffffeu;
                  }
                }
                if (!hslot) { if (deep) { ++nhit; ++w_hits; } }
                else {
                  unsigned long long coff = 0;
                  if (lane == 0) coff = atomicAdd(bv.cigar_used, (unsigned long long)nm);
                  coff = rl64(coff, 0);
                  if (coff + nm <= bv.cigar_cap)
                    for (uint32_t r = lane; r < nm; r += 64) bv.cigar_pool[coff + r] = xl32(&mruns[r]);
                  if (lane == 0) {
                    ugs_hit *h = nhit < lv.hit_slots ? &bv.hits[(uint64_t)unit * lv.hit_slots + nhit]
                                                     : &bv.xpool[(uint64_t)xcur * UGS_XBLOCK + (nhit - lv.hit_slots) % UGS_XBLOCK];
                    h->query = qi; h->target = t; h->ids = ids; h->mism = mcols - ids; h->gaps_int = gaps; h->aln_len = cols;
                    h->opens = opens; h->qlo = Loi; h->qhi = Loi + Leni - 1; h->tlo = Loj; h->thi = Loj + Lenj - 1; h->ql = QL; h->tl = TL;
                    h->strand = strand; h->cigar_off = coff; h->cigar_len = nm; h->cols = cols;
                    h->raw_score = (float)score2 * 0.5f; h->flags = UGS_HIT_LOCAL;
                  }
                  ++nhit; ++w_hits;
                }
              }
              wave_sync();
              tl3 += ACLK() - tq;
              // localmulti.cpp:104-110: the walk resumes behind the HSP
              const uint32_t NewT = Loj + Lenj;                                           // GetHij() + 1
              cur = NewT > tt ? NewT : tt + 1;
            }
          }
          curT = max(cur, tend);
        }
      }
      // Terminator::Terminate (terminator.cpp:64-100): one accept or reject per target
      if (any_accept) ++nacc; else ++nrej;
      if (nacc == (deep ? (uint32_t)db.acc_limit : max_acc) || nrej == max_rej) { walk_ended = true; break; }
      wave_sync();
    }
    if (deep) {
      if (cont) {
        if (!walk_ended && page + ncand < n_total) { page += ncand; ncand = n_total - page < 64u ? n_total - page : 64u; wave_sync(); goto next_page; }
        if (lane == 0) bv.walk_state[unit].xhead = xhead;         // (only this: the parked counters stay, the pass can be run again)
      } else if (!walk_ended && ncand == K && lane == 0) {
        // the walk used up a FULL list of K candidates without meeting a limit: parked for the continuation pass
        UgsWalkState st; st.nacc = nacc; st.nrej = nrej; st.nvis = ncand; st.xhead = 0xffffffffu; st.xcur = 0xffffffffu; st.pad0 = nhit; st.pad1 = st.pad2 = 0;
        bv.walk_state[unit] = st;
        bv.open_list[atomicAdd(&ctr[UGS_CTR_OPEN], 1ull)] = unit;
      }
    }
    if (lane == 0) bv.hit_n[unit] = nhit;
    wave_sync();
  }
  if (lane == 0) {
    atomicAdd(&ctr[UGS_CTR_TLETTERS], w_tletters); atomicAdd(&ctr[UGS_CTR_PAIRS], w_pairs);
    atomicAdd(&ctr[UGS_CTR_CELLS], cells); atomicAdd(&ctr[UGS_CTR_HITS], w_hits);
  }
  if (tid == 0) { atomicAdd(&ctr[UGS_CTR_T4], tl0); atomicAdd(&ctr[UGS_CTR_T5], tl1); atomicAdd(&ctr[UGS_CTR_T6], tl2); atomicAdd(&ctr[UGS_CTR_T7], tl3); }
}

size_t ugs_local_wave_lds(uint32_t W, uint32_t max_qlen, uint32_t max_tlen, uint32_t seed_cap)
{
  const size_t maxq = (max_qlen + 15u) & ~15u, maxt = (max_tlen + 15u) & ~15u;
  size_t q2 = 64; while (q2 < maxq) q2 <<= 1;
  return ((((size_t)W * 13 + 15) & ~(size_t)15) + maxq * 2 + q2 * 4 + maxt + (size_t)seed_cap * 4 + LOC_MAXARS * 16 + 15) & ~(size_t)15;
}

int ugs_local_blocks_per_cu(int threads, size_t lds)
{
  int n = 0;
  if (hipFuncSetAttribute((const void *)k_local, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k_local, threads, lds) != hipSuccess || n < 1) n = 1;
  return n;
}

int ugs_launch_local(const UgsDbView &db, const UgsBatchView &b, const UgsLocalView &lv, int grid, int wpb, size_t lds, hipStream_t st)
{
  const uint32_t wave_lds = (uint32_t)((lds - 2560) / wpb);
  HIPCHK(hipFuncSetAttribute((const void *)k_local, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_local, dim3(grid), dim3(64 * wpb), lds, st, db, b, lv, wave_lds);
  HIPCHK(hipGetLastError());
  return UGS_OK;
}
