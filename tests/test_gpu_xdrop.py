"""GPU parity for the gapped x-drop rows: ugs_xdrop_batch (HIP, through the C-ABI)
against (1) the reference's own answers committed under tests/golden/xdrop_*.txt and (2) the oracle
on seeded batches, bit-exact: score, HSP coordinates and the full path."""
import numpy as np
import pytest

import golden_util
import orc
from usearch12_amd import capi
from usearch12_amd.abi import XDROP_ALIGN, XDROP_FWD, XDROP_BWD, XDROP_JOB_DTYPE, path_text

pytestmark = pytest.mark.gpu
MODE = {"A": XDROP_ALIGN, "F": XDROP_FWD, "B": XDROP_BWD}


def pack(seqs):
    offs = np.zeros(len(seqs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    return np.frombuffer("".join(seqs).encode(), np.uint8).copy(), offs


def run_cases(is_nucleo, x, cases, **pk):
    """cases: list of (mode, a, b, anc) -> list of (score, loi, loj, leni, lenj, path)"""
    jobs = np.zeros(len(cases), XDROP_JOB_DTYPE)
    for k, (mode, a, b, anc) in enumerate(cases):
        jobs[k] = (k, k, anc[0], anc[1], anc[2], mode)
    p = capi.xdrop_params(is_nucleo, xdrop=x, **pk)
    hsps, pool = capi.xdrop_batch(p, pack([c[1] for c in cases]), pack([c[2] for c in cases]), jobs)
    out = []
    for h in hsps:
        out.append((float(h["score"]), int(h["loi"]), int(h["loj"]), int(h["leni"]), int(h["lenj"]),
                    path_text(pool, h["path_off"], h["path_len"])))
    return out


@pytest.mark.parametrize("name", ["nt", "aa"])
def test_xdrop_reference_answers(name):
    cases = golden_util.load_xdrop(name)
    bad = []
    for x in sorted(set(c["x"] for c in cases)):
        sub = [c for c in cases if c["x"] == x]
        got = run_cases(name == "nt", x, [(MODE[c["mode"]], c["a"], c["b"], c["anc"]) for c in sub])
        for c, g in zip(sub, got):
            w = c["want"]
            if c["mode"] == "A":
                ok = g == w
            else:
                ok = (g[0], g[3], g[4], g[5]) == (w[0], w[3], w[4], w[5])
            if not ok:
                bad.append((c["mode"], x, len(c["a"]), len(c["b"]), c["anc"], g[:5], w[:5]))
    assert not bad, (len(bad), bad[:5])


def test_xdrop_reference_kat():
    got = run_cases(False, 32.0, [(XDROP_FWD, "SEQVENCE", "SEQVECE", (0, 0, 0))])[0]
    assert (got[0], got[3], got[4], got[5]) == (27.0, 8, 7, "MMMMMDMM")


def _random_batch(seed, aa, n, lmin, lmax):
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("mgx", os.path.join(golden_util.GOLD, "make_golden_xdrop.py"))
    mgx = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mgx)
    rng = np.random.default_rng(seed)
    alpha = mgx.AA if aa else mgx.NT
    cases = []
    for k in range(n):
        L = int(rng.integers(lmin, lmax))
        a = mgx.rand_seq(rng, L, alpha) if k % 5 else mgx.low_complexity(rng, L, alpha)
        b = mgx.mutate(rng, a, alpha, *[(0.02, 0.005, 0.005), (0.08, 0.03, 0.03), (0.2, 0.05, 0.05)][k % 3], burst=0.3 * (k % 2))
        if k % 4 == 0:
            b = b[: max(1, len(b) // 2)] + mgx.rand_seq(rng, int(rng.integers(1, 200)), alpha)
        anc = mgx.find_anchor(rng, a, b, 4 if aa else 8)
        if anc is not None and k % 3:
            cases.append((XDROP_ALIGN, a, b, anc))
        else:
            cases.append((XDROP_FWD if k % 2 else XDROP_BWD, a, b, (0, 0, 0)))
    return cases


@pytest.mark.parametrize("aa,x,seed", [(False, 32.0, 1), (False, 12.0, 2), (True, 32.0, 3), (True, 60.0, 4), (False, 200.0, 5)])
def test_xdrop_matches_oracle(aa, x, seed):
    cases = _random_batch(seed, aa, 1500, 5, 900)
    got = run_cases(not aa, x, cases)
    p = orc.xdrop_params(not aa, x)
    bad = []
    for (mode, a, b, anc), g in zip(cases, got):
        o = orc.xdrop_job(p, a, b, mode, anc)
        want = o[:6] if mode != XDROP_FWD else (o[0], 0, 0, o[3], o[4], o[5])
        if g != want:
            bad.append((mode, len(a), len(b), anc, g[:5], want[:5]))
    assert not bad, (len(bad), bad[:5])
    ms, cells = capi.xdrop_last_stats()
    assert ms > 0 and cells > 0


def test_xdrop_wide_window_and_retry():
    """X large enough that the live window spans several 64-lane chunks and the per-wave traceback scratch of the
    first pass overflows (the job is redone with worst-case scratch)."""
    rng = np.random.default_rng(7)
    a = "".join("ACGT"[i] for i in rng.integers(0, 4, 3000))
    b = a[:1500] + "".join("ACGT"[i] for i in rng.integers(0, 4, 40)) + a[1500:]
    cases = [(XDROP_FWD, a, b, (0, 0, 0)), (XDROP_BWD, a, b, (0, 0, 0)), (XDROP_ALIGN, a, b, (700, 700, 12))]
    got = run_cases(True, 2000.0, cases)
    p = orc.xdrop_params(True, 2000.0)
    for (mode, a_, b_, anc), g in zip(cases, got):
        o = orc.xdrop_job(p, a_, b_, mode, anc)
        want = o[:6] if mode != XDROP_FWD else (o[0], 0, 0, o[3], o[4], o[5])
        assert g == want, (mode, g[:5], want[:5])


def test_xdrop_bad_arguments():
    p = capi.xdrop_params(True)
    jobs = np.zeros(1, XDROP_JOB_DTYPE)
    jobs[0] = (0, 0, 5, 5, 10, XDROP_ALIGN)               # anchor runs off the end
    with pytest.raises(capi.UgsError):
        capi.xdrop_batch(p, pack(["ACGTACGT"]), pack(["ACGTACGT"]), jobs)
    jobs[0] = (0, 3, 0, 0, 0, XDROP_FWD)                  # sequence index out of range
    with pytest.raises(capi.UgsError):
        capi.xdrop_batch(p, pack(["ACGTACGT"]), pack(["ACGTACGT"]), jobs)


@pytest.mark.parametrize("aa,x,pk", [(False, 20.5, dict(local_open=-5.0, local_ext=-2.0, mismatch=-3.0)),
                                      (False, 7.0, dict(local_open=-2.5, local_ext=-0.5, match=2.0, mismatch=-1.5)),
                                      (True, 40.0, dict(local_open=-11.0, local_ext=-1.0))])
def test_xdrop_non_default_scoring(aa, x, pk):
    """gap penalties / nt scores / X other than the defaults (all multiples of 0.5, as the integer DP requires)"""
    cases = _random_batch(11, aa, 600, 5, 500)
    got = run_cases(not aa, x, cases, **pk)
    p = orc.xdrop_params(not aa, x)
    for k, v in pk.items():
        setattr(p, k, v)
    bad = []
    for (mode, a, b, anc), g in zip(cases, got):
        o = orc.xdrop_job(p, a, b, mode, anc)
        want = o[:6] if mode != XDROP_FWD else (o[0], 0, 0, o[3], o[4], o[5])
        if g != want:
            bad.append((mode, len(a), len(b), anc, g[:5], want[:5]))
    assert not bad, (len(bad), bad[:5])


def _raw_batch(cases, is_nucleo, x):
    jobs = np.zeros(len(cases), XDROP_JOB_DTYPE)
    for k, (mode, a, b, anc) in enumerate(cases):
        jobs[k] = (k, k, anc[0], anc[1], anc[2], mode)
    return capi.xdrop_params(is_nucleo, xdrop=x), pack([c[1] for c in cases]), pack([c[2] for c in cases]), jobs


@pytest.mark.parametrize("aa,x,seed", [(False, 200.0, 5), (True, 60.0, 4)])
def test_xdrop_same_batch_fifty_times(aa, x, seed):
    """Determinism: the batch that was red in GPUTEST_r02 (seed 5, X = 200: windows of several
    64-column chunks, long tracebacks) through k_xdrop 50 times in one process - every run's device output (HSP
    records and the run pool, compared as raw bytes) equals the first run's, and the first equals the oracle."""
    cases = _random_batch(seed, aa, 1500, 5, 900)
    p, A, B, jobs = _raw_batch(cases, not aa, x)
    first = None
    for it in range(50):
        hsps, pool = capi.xdrop_batch(p, A, B, jobs)
        raw = (hsps.tobytes(), pool.tobytes())
        if first is None:
            first = raw
            op = orc.xdrop_params(not aa, x)
            for (mode, a, b, anc), h in zip(cases, hsps):
                o = orc.xdrop_job(op, a, b, mode, anc)
                want = o[:6] if mode != XDROP_FWD else (o[0], 0, 0, o[3], o[4], o[5])
                got = (float(h["score"]), int(h["loi"]), int(h["loj"]), int(h["leni"]), int(h["lenj"]),
                       path_text(pool, h["path_off"], h["path_len"]))
                assert got == want, (mode, len(a), len(b), anc, got[:5], want[:5])
        else:
            assert raw == first, "run %d differs from run 0" % it
