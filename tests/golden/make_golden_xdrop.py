#!/usr/bin/env python3
"""Known answers for the gapped x-drop rows, produced by the UNMODIFIED
reference: oracle/_ref/ref_xdrop is oracle/ref_xdrop_main.cpp (our driver) linked with the
reference's own objects by oracle/build_ref.sh; it calls XDropFwdFastMem / XDropBwdFastMem /
XDropAlignMem directly.  Runs only where /root/reference exists.

Writes tests/golden/xdrop_{nt,aa}.txt: two lines per case,
   case line    : mode X A B [anc_loi anc_loj anc_len]        (exactly what ref_xdrop read)
   answer line  : = score leni lenj path | = score loi loj leni lenj path
"""
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_xdrop")
NT = "ACGT"
AA = "ACDEFGHIKLMNPQRSTVWY"


def rand_seq(rng, n, alpha):
    return "".join(alpha[i] for i in rng.integers(0, len(alpha), n))


def mutate(rng, s, alpha, p_sub, p_del, p_ins, burst=0.0):
    out = []
    i = 0
    while i < len(s):
        r = rng.random()
        if r < p_sub:
            out.append(alpha[rng.integers(0, len(alpha))])
        elif r < p_sub + p_del:
            if burst and rng.random() < burst:
                i += int(rng.integers(1, 12))
        elif r < p_sub + p_del + p_ins:
            k = int(rng.integers(1, 12)) if burst and rng.random() < burst else 1
            out.append(rand_seq(rng, k, alpha))
            out.append(s[i])
        else:
            out.append(s[i])
        i += 1
    return "".join(out) or alpha[0]


def low_complexity(rng, n, alpha):
    unit = rand_seq(rng, int(rng.integers(1, 5)), alpha)
    s = (unit * (n // len(unit) + 1))[:n]
    return mutate(rng, s, alpha, 0.03, 0.01, 0.01)


def find_anchor(rng, a, b, k):
    """an exact k-mer of a (random start) that occurs in b, extended to the right"""
    idx = {}
    for j in range(len(b) - k + 1):
        idx.setdefault(b[j:j + k], j)
    starts = list(range(0, len(a) - k + 1))
    rng.shuffle(starts)
    for i in starts[:200]:
        j = idx.get(a[i:i + k])
        if j is not None:
            n = k
            while i + n < len(a) and j + n < len(b) and a[i + n] == b[j + n] and n < 40:
                n += 1
            return i, j, n
    return None


def cases(aa, seed):
    rng = np.random.default_rng(seed)
    alpha = AA if aa else NT
    k = 4 if aa else 8
    out = []
    xs = [32, 32, 32, 16, 8, 64, 100, 5]
    # tiny and degenerate shapes
    for la, lb in [(1, 1), (1, 5), (5, 1), (2, 2), (2, 3), (3, 2), (3, 3)]:
        for _ in range(3):
            a = rand_seq(rng, la, alpha)
            b = a[:lb] if rng.random() < 0.5 and la >= lb else rand_seq(rng, lb, alpha)
            out.append(("F", 32, a, b))
            out.append(("B", 32, a, b))
    for n in range(150):
        L = int(rng.integers(8, 700))
        a = rand_seq(rng, L, alpha) if n % 7 else low_complexity(rng, L, alpha)
        kind = n % 5
        if kind == 0:
            b = mutate(rng, a, alpha, 0.02, 0.005, 0.005)
        elif kind == 1:
            b = mutate(rng, a, alpha, 0.10, 0.03, 0.03)
        elif kind == 2:
            b = mutate(rng, a, alpha, 0.04, 0.02, 0.02, burst=0.5)
        elif kind == 3:     # related prefix, unrelated tail: the x-drop has to stop
            cut = int(rng.integers(1, L))
            b = mutate(rng, a[:cut], alpha, 0.03, 0.01, 0.01) + rand_seq(rng, int(rng.integers(1, 300)), alpha)
        else:
            b = mutate(rng, a, alpha, 0.25, 0.05, 0.05)
        if n % 11 == 0:     # soft-masked / wildcard letters score 0 or as their upper case
            pos = int(rng.integers(0, len(b)))
            b = b[:pos] + b[pos:pos + 9].lower() + b[pos + 9:]
            pos = int(rng.integers(0, len(a)))
            a = a[:pos] + ("X" if aa else "N") + a[pos + 1:]
        x = xs[n % len(xs)]
        out.append(("F", x, a, b))
        out.append(("B", x, a[::-1], b[::-1]) if n % 2 else ("B", x, a, b))
        anc = find_anchor(rng, a, b, k)
        if anc:
            out.append(("A", x, a, b) + anc)
            if anc[2] > 2:                      # shorter anchors incl. the AncLen<=1 early-out
                out.append(("A", x, a, b, anc[0], anc[1], int(rng.integers(0, 3))))
    # sides longer than g_MaxL=4096 go through the split drivers
    for la, shape in [(9500, 0), (5200, 1), (12500, 2)]:
        a = rand_seq(rng, la, alpha)
        b = mutate(rng, a, alpha, 0.03, 0.004, 0.004)
        if shape == 1:
            b = b[: len(b) - 300] + rand_seq(rng, 400, alpha)
        anc = None
        while anc is None:
            anc = find_anchor(rng, a, b, k + 4)
        if shape == 2:      # anchor near the start: only the forward side splits
            i0 = 100
            j0 = b.find(a[i0:i0 + 14])
            if j0 >= 0:
                anc = (i0, j0, 14)
        out.append(("A", 32, a, b) + anc)
    return out


def main():
    assert os.path.exists(REF), "build the reference first: oracle/build_ref.sh"
    for aa in (False, True):
        name = "aa" if aa else "nt"
        cs = cases(aa, 4242 + aa)
        text = "".join(" ".join(str(x) for x in c) + "\n" for c in cs)
        res = subprocess.run([REF, name], input=text.encode(), stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
        assert len(res) == len(cs), (len(res), len(cs))
        with open(os.path.join(HERE, "xdrop_%s.txt" % name), "w") as f:
            for c, r in zip(cs, res):
                f.write(" ".join(str(x) for x in c) + "\n= " + r + "\n")
        print(name, len(cs), "cases")


if __name__ == "__main__":
    main()
