"""Seeded synthetic FASTA-shaped inputs for the usearch_global hot path.

Workload model follows SURVEY.md section 8(d) / BASELINE.md section 3:
  nt:  DB = N x L iid uniform ACGT; queries = 90 % random DB member mutated with
       per-base 1.5 % substitution, 0.15 % deletion, 0.15 % insertion, 10 % iid random.
  aa:  DB drawn from Robinson-Robinson background; queries 10 % substitution,
       0.5 % indels (0.25 % del + 0.25 % ins), 10 % unrelated random.
Sequences are produced as one concatenated uint8 ASCII buffer + uint64 offsets
(the layout the C-ABI takes), and can be written out as FASTA for the reference
binary.  Pure numpy, vectorised, deterministic for a given (seed, sizes).
"""
import numpy as np

NT = np.frombuffer(b"ACGT", dtype=np.uint8)
AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
# Robinson & Robinson (1991) amino-acid background frequencies, order ACDEFGHIKLMNPQRSTVWY
RR_FREQ = np.array([0.07805, 0.01925, 0.05364, 0.06295, 0.03856, 0.07377, 0.02199,
                    0.05142, 0.05744, 0.09019, 0.02243, 0.04487, 0.05203, 0.04264,
                    0.05129, 0.07120, 0.05841, 0.06441, 0.01330, 0.03216])
RR_FREQ = RR_FREQ / RR_FREQ.sum()


class SeqSet:
    """Concatenated sequences: `seqs` uint8 ASCII, `offs` uint64 [n+1], labels lazily built."""

    def __init__(self, seqs, offs, label_fn):
        self.seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        self.offs = np.ascontiguousarray(offs, dtype=np.uint64)
        self.n = len(offs) - 1
        self._label_fn = label_fn

    def label(self, i):
        return self._label_fn(i)

    def labels(self):
        return [self._label_fn(i) for i in range(self.n)]

    def seq(self, i):
        return self.seqs[int(self.offs[i]):int(self.offs[i + 1])].tobytes()

    def slice(self, lo, hi):
        o = self.offs[lo:hi + 1]
        base = int(o[0])
        f = self._label_fn
        return SeqSet(self.seqs[base:int(o[-1])], o - np.uint64(base), lambda i, lo=lo: f(i + lo))

    def write_fasta(self, path):
        with open(path, "wb") as f:
            # chunked, vectorised: header bytes are built per record but sequence bytes copied raw
            mv = memoryview(self.seqs)
            offs = self.offs
            buf = []
            for i in range(self.n):
                buf.append(b">" + self._label_fn(i).encode() + b"\n")
                buf.append(mv[int(offs[i]):int(offs[i + 1])])
                buf.append(b"\n")
                if len(buf) >= 30000:
                    f.write(b"".join(buf))
                    buf = []
            f.write(b"".join(buf))


def _random_letters(rng, n, alphabet, freq=None):
    if freq is None:
        return alphabet[rng.integers(0, len(alphabet), size=n, dtype=np.uint8)]
    cdf = np.cumsum(freq)
    cdf[-1] = 1.0
    return alphabet[np.searchsorted(cdf, rng.random(n), side="right").astype(np.uint8)]


def make_db(seed, n, length, aa=False):
    rng = np.random.default_rng([seed, 0xDB])
    alphabet, freq = (AA, RR_FREQ) if aa else (NT, None)
    seqs = _random_letters(rng, n * length, alphabet, freq)
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(length)
    return SeqSet(seqs, offs, lambda i: "t%d" % i)


def make_queries(seed, db, n, length, aa=False, p_sub=None, p_del=None, p_ins=None,
                 frac_random=0.10):
    """Mutated copies of random DB members + unrelated random sequences."""
    rng = np.random.default_rng([seed, 0x51])
    alphabet, freq = (AA, RR_FREQ) if aa else (NT, None)
    if p_sub is None:
        p_sub, p_del, p_ins = (0.10, 0.0025, 0.0025) if aa else (0.015, 0.0015, 0.0015)
    is_rand = rng.random(n) < frac_random
    src = rng.integers(0, db.n, size=n)
    # gather source letters (all DB members of a synthetic DB have equal length)
    dlen = int(db.offs[1] - db.offs[0])
    assert np.all(np.diff(db.offs.astype(np.int64)) == dlen)
    base = db.seqs.reshape(db.n, dlen)[src].copy()                     # [n, dlen]
    rnd_rows = np.flatnonzero(is_rand)
    if len(rnd_rows):
        r = _random_letters(rng, len(rnd_rows) * dlen, alphabet, freq).reshape(-1, dlen)
        base[rnd_rows, :] = r[:, :dlen]
    u = rng.random((n, dlen))
    u[rnd_rows, :] = 1.0                                               # no mutation on random rows
    sub = u < p_sub
    dele = (u >= p_sub) & (u < p_sub + p_del)
    ins = (u >= p_sub + p_del) & (u < p_sub + p_del + p_ins)
    # substitutions: replace by a *different* letter
    nsub = int(sub.sum())
    if nsub:
        old = base[sub]
        k = len(alphabet)
        old_idx = np.searchsorted(alphabet, old)   # alphabets are sorted ASCII
        new_idx = (old_idx + rng.integers(1, k, size=nsub)) % k
        base[sub] = alphabet[new_idx]
    rep = np.ones((n, dlen), dtype=np.int64)
    rep[dele] = 0
    rep[ins] = 2
    flat = np.repeat(base.reshape(-1), rep.reshape(-1))
    lens = rep.sum(axis=1)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens).astype(np.uint64)
    # the second copy of an "ins" base becomes a fresh random letter
    start = np.cumsum(rep.reshape(-1)) - rep.reshape(-1)
    ins_pos = start[ins.reshape(-1)] + 1
    if len(ins_pos):
        flat[ins_pos] = _random_letters(rng, len(ins_pos), alphabet, freq)
    src_l = src.copy()
    israndl = is_rand.copy()

    def label(i):
        return "q%d;src=%s" % (i, "rand" if israndl[i] else "t%d" % src_l[i])
    qs = SeqSet(flat, offs, label)
    qs.src = np.where(is_rand, -1, src)
    return qs


def revcomp_some(seed, qs, frac=0.5):
    """Reverse-complement a random subset of nt queries (for -strand both tests)."""
    rng = np.random.default_rng([seed, 0x7C])
    comp = np.arange(256, dtype=np.uint8)           # unknown letters stay as they are
    for a, b in zip(b"ABCDGHKMNRSTUVWXY", b"TVGHCDMKNYSAABWXR"):
        comp[a] = b
        comp[a | 0x20] = b | 0x20
    seqs = qs.seqs.copy()
    flip = rng.random(qs.n) < frac
    for i in np.flatnonzero(flip):
        lo, hi = int(qs.offs[i]), int(qs.offs[i + 1])
        seqs[lo:hi] = comp[seqs[lo:hi]][::-1]
    out = SeqSet(seqs, qs.offs, qs._label_fn)
    out.flip = flip
    return out


def make_reads(seed, n_reads, n_species=None, length=300, p_sub=0.01, p_del=0.001, p_ins=0.001, alpha=1.2,
               chunk=200_000, dup_frac=0.0):
    """cluster_fast workload: reads drawn from `n_species` random roots with Pareto(alpha)
    abundances, each read = its root with per-base 1 % substitutions and 0.1 % / 0.1 % deletions / insertions.
    dup_frac > 0 makes that share of the reads exact copies of an earlier read (dereplication fodder beyond the
    error-free reads the model yields by itself).  Labels r<i>;sp=<species>.  Built in chunks (5 M x 300 fits)."""
    rng = np.random.default_rng([seed, 0xC3])
    if n_species is None:
        n_species = max(1, n_reads // 100)
    roots = _random_letters(rng, n_species * length, NT).reshape(n_species, length)
    w = (1.0 - rng.random(n_species)) ** (-1.0 / alpha)
    cdf = np.cumsum(w / w.sum())
    cdf[-1] = 1.0
    sp = np.searchsorted(cdf, rng.random(n_reads), side="right").astype(np.int64)
    parts, lens_all = [], []
    for lo in range(0, n_reads, chunk):
        hi = min(n_reads, lo + chunk)
        n = hi - lo
        base = roots[sp[lo:hi]].copy()
        u = rng.random((n, length), dtype=np.float32)
        sub = u < p_sub
        dele = (u >= p_sub) & (u < p_sub + p_del)
        ins = (u >= p_sub + p_del) & (u < p_sub + p_del + p_ins)
        nsub = int(sub.sum())
        if nsub:
            old_idx = np.searchsorted(NT, base[sub])
            base[sub] = NT[(old_idx + rng.integers(1, 4, size=nsub)) % 4]
        rep = np.ones((n, length), dtype=np.int8)
        rep[dele] = 0
        rep[ins] = 2
        repf = rep.reshape(-1).astype(np.int64)
        flat = np.repeat(base.reshape(-1), repf)
        start = np.cumsum(repf) - repf
        ins_pos = start[ins.reshape(-1)] + 1
        if len(ins_pos):
            flat[ins_pos] = _random_letters(rng, len(ins_pos), NT)
        parts.append(flat)
        lens_all.append(rep.sum(axis=1, dtype=np.int64))
    lens = np.concatenate(lens_all)
    offs = np.zeros(n_reads + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens).astype(np.uint64)
    seqs = np.concatenate(parts)
    if dup_frac > 0 and n_reads > 1:
        # exact copies of an earlier read (same length only keeps the layout: copy letters when lengths agree)
        dst = np.flatnonzero(rng.random(n_reads) < dup_frac)
        dst = dst[dst > 0]
        srcs = (rng.random(len(dst)) * dst).astype(np.int64)
        for d, s0 in zip(dst, srcs):
            if lens[d] == lens[s0]:
                seqs[int(offs[d]):int(offs[d + 1])] = seqs[int(offs[s0]):int(offs[s0 + 1])]
    spl = sp
    rs = SeqSet(seqs, offs, lambda i: "r%d;sp=%d" % (i, spl[i]))
    rs.species = sp
    return rs


CONFIGS = {
    # name: (seed, db_n, n_queries, length, aa, id)
    "C1": (1, 50_000, 10_000, 250, False, 0.97),
    "C2": (2, 1_000_000, 1_000_000, 250, False, 0.97),
    "C4": (4, 5_000_000, 10_000_000, 250, False, 0.97),
    "C5": (5, 2_000_000, 1_000_000, 300, True, 0.80),
}


def make_config(name, db_n=None, n_queries=None):
    seed, dn, qn, length, aa, ident = CONFIGS[name]
    dn = db_n or dn
    qn = n_queries or qn
    db = make_db(seed, dn, length, aa)
    qs = make_queries(seed, db, qn, length, aa)
    return db, qs, ident


def make_hard(seed, n_fam, fam_size, n_queries, lmin=150, lmax=400, aa=False):
    """Adversarial parity workload: DB of sequence *families* (members 1-10 % diverged from a
    family root, so queries have several near-equal candidates and the prevMax/2 cutoff and
    tie orders matter), variable lengths, low-complexity runs (mask-relevant), ambiguity
    codes and lower-case stretches in queries."""
    rng = np.random.default_rng([seed, 0x4A])
    alphabet, freq = (AA, RR_FREQ) if aa else (NT, None)
    k = len(alphabet)

    def mutate(s, p_sub, p_indel):
        s = s.copy()
        u = rng.random(len(s))
        sub = u < p_sub
        if sub.any():
            s[sub] = alphabet[rng.integers(0, k, size=int(sub.sum()))]
        keep = ~((u >= p_sub) & (u < p_sub + p_indel))
        s = s[keep]
        u2 = rng.random(len(s))
        insp = np.flatnonzero(u2 < p_indel)
        if len(insp):
            s = np.insert(s, insp, alphabet[rng.integers(0, k, size=len(insp))])
        return s

    def lowcomplex(s):
        s = s.copy()
        for _ in range(int(rng.integers(0, 3))):
            pos = int(rng.integers(0, max(1, len(s) - 20)))
            n = int(rng.integers(4, 14))
            if rng.random() < 0.5:
                s[pos:pos + n] = alphabet[rng.integers(0, k)]
            else:
                a, b = alphabet[rng.integers(0, k, size=2)]
                seg = np.tile(np.array([a, b], dtype=np.uint8), n)[:len(s[pos:pos + 2 * n])]
                s[pos:pos + len(seg)] = seg
        return s

    dbs = []
    for f in range(n_fam):
        L = int(rng.integers(lmin, lmax + 1))
        root = lowcomplex(_random_letters(rng, L, alphabet, freq))
        for m in range(fam_size):
            dbs.append(mutate(root, rng.uniform(0.0, 0.10), rng.uniform(0.0, 0.01)) if m else root)
    perm = rng.permutation(len(dbs))
    dbs = [dbs[i] for i in perm]
    doffs = np.zeros(len(dbs) + 1, dtype=np.uint64)
    doffs[1:] = np.cumsum([len(s) for s in dbs])
    db = SeqSet(np.concatenate(dbs), doffs, lambda i: "t%d" % i)

    qs, src = [], []
    wild = np.frombuffer(b"XBZ" if aa else b"NRYKMSWN", dtype=np.uint8)
    for q in range(n_queries):
        r = rng.random()
        if r < 0.08:
            L = int(rng.integers(lmin, lmax + 1))
            s = _random_letters(rng, L, alphabet, freq); src.append(-1)
        else:
            t = int(rng.integers(0, db.n))
            base = db.seqs[int(db.offs[t]):int(db.offs[t + 1])]
            s = mutate(base, rng.uniform(0.0, 0.05), rng.uniform(0.0, 0.006)); src.append(t)
            if rng.random() < 0.15 and len(s) > 60:          # truncated ends -> terminal gaps
                a = int(rng.integers(0, 25)); b = int(rng.integers(0, 25))
                s = s[a:len(s) - b]
        if rng.random() < 0.2 and len(s) > 10:               # ambiguity codes
            pos = rng.integers(0, len(s), size=int(rng.integers(1, 4)))
            s = s.copy(); s[pos] = wild[rng.integers(0, len(wild), size=len(pos))]
        if rng.random() < 0.15 and len(s) > 30:              # lower-case stretch
            a = int(rng.integers(0, len(s) - 20)); s = s.copy()
            s[a:a + int(rng.integers(3, 20))] |= 0x20
        qs.append(s)
    qoffs = np.zeros(len(qs) + 1, dtype=np.uint64)
    qoffs[1:] = np.cumsum([len(s) for s in qs])
    srcl = list(src)
    qset = SeqSet(np.concatenate(qs), qoffs, lambda i: "q%d;src=%s" % (i, "rand" if srcl[i] < 0 else "t%d" % srcl[i]))
    return db, qset


def make_local_queries(seed, db, n_queries, aa=False):
    """usearch_local parity workload on top of an existing DB (normally make_hard's families): fragments of a
    target between unrelated flanks, two fragments of one target split by junk (several HSPs on one target),
    chimeras of two targets, repeats of one fragment, distant homologs near the e-value gate, unrelated
    sequences and very short queries; some lower-case stretches and ambiguity codes."""
    rng = np.random.default_rng([seed, 0x10CA])
    alphabet, freq = (AA, RR_FREQ) if aa else (NT, None)
    k = len(alphabet)
    wild = np.frombuffer(b"XBZ" if aa else b"NRYKMSWN", dtype=np.uint8)

    def rnd(n):
        return _random_letters(rng, int(n), alphabet, freq)

    def mutate(s, p_sub, p_indel):
        s = s.copy()
        u = rng.random(len(s))
        sub = u < p_sub
        if sub.any():
            s[sub] = alphabet[rng.integers(0, k, size=int(sub.sum()))]
        s = s[~((u >= p_sub) & (u < p_sub + p_indel))]
        insp = np.flatnonzero(rng.random(len(s)) < p_indel)
        if len(insp):
            s = np.insert(s, insp, alphabet[rng.integers(0, k, size=len(insp))])
        return s

    def target(t):
        return db.seqs[int(db.offs[t]):int(db.offs[t + 1])]

    def frag(t, fmin=0.3, fmax=1.0):
        s = target(t)
        n = max(12, int(len(s) * rng.uniform(fmin, fmax)))
        a = int(rng.integers(0, max(1, len(s) - n + 1)))
        return s[a:a + n]

    qs, src = [], []
    for q in range(n_queries):
        r = rng.random()
        t = int(rng.integers(0, db.n))
        div = (rng.uniform(0.0, 0.12), rng.uniform(0.0, 0.01))
        if r < 0.08:
            s = rnd(rng.integers(40, 400)); t = -1
        elif r < 0.30:
            s = np.concatenate([rnd(rng.integers(0, 80)), mutate(frag(t), *div), rnd(rng.integers(0, 80))])
        elif r < 0.50:      # two pieces of the same target, junk between them
            base = target(t)
            cut = int(len(base) * rng.uniform(0.3, 0.7))
            s = np.concatenate([mutate(base[:cut], *div), rnd(rng.integers(40, 160)), mutate(base[cut:], *div)])
        elif r < 0.62:      # chimera
            t2 = int(rng.integers(0, db.n))
            s = np.concatenate([mutate(frag(t, 0.3, 0.6), *div), mutate(frag(t2, 0.3, 0.6), *div)])
        elif r < 0.70:      # the same fragment twice
            f = mutate(frag(t, 0.2, 0.5), *div)
            s = np.concatenate([f, rnd(rng.integers(5, 40)), mutate(f, 0.02, 0.002)])
        elif r < 0.85:      # distant
            s = mutate(target(t), rng.uniform(0.15, 0.45 if aa else 0.30), rng.uniform(0.0, 0.03))
        elif r < 0.90:      # very short
            s = mutate(frag(t, 0.02, 0.12), 0.02, 0.0)[: int(rng.integers(1, 30))]
        else:
            s = mutate(target(t), *div)
        if len(s) == 0:
            s = rnd(1)
        s = s.copy()
        if rng.random() < 0.2 and len(s) > 10:
            pos = rng.integers(0, len(s), size=int(rng.integers(1, 4)))
            s[pos] = wild[rng.integers(0, len(wild), size=len(pos))]
        if rng.random() < 0.15 and len(s) > 30:
            a = int(rng.integers(0, len(s) - 20)); b = a + int(rng.integers(5, 20))
            s[a:b] = np.char.lower(s[a:b].view("S1")).view(np.uint8)
        qs.append(s); src.append(t)
    qoffs = np.zeros(len(qs) + 1, dtype=np.uint64)
    qoffs[1:] = np.cumsum([len(s) for s in qs])
    srcl = list(src)
    out = SeqSet(np.concatenate(qs), qoffs, lambda i: "q%d;src=%s" % (i, "rand" if srcl[i] < 0 else "t%d" % srcl[i]))
    out.src = np.array(src)
    return out
