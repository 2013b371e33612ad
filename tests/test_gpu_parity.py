"""GPU parity tests proper (-m gpu): the HIP path through the C-ABI vs (a) the reference's
golden text and (b) the oracle, bit-exact on hit identity, coordinates, paths and text."""
import os

import numpy as np
import pytest

import golden_util as G
import orc
from usearch12_amd import capi

pytestmark = pytest.mark.gpu

# Round-1 device scope: the Big ranking path (DB > -big) for nt and aa; the small path is
# implemented too and covered by the same cases.
CASES = G.case_names()


def _run_gpu(c, db, qs):
    p = capi.params(is_nucleo=not c["aa"], id=c["id"], **G.params_kw(c))
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    hits, nh, pool = gdb.search(qs.seqs, qs.offs)
    return p, gdb, hits, nh, pool


@pytest.mark.parametrize("name", CASES)
def test_gpu_matches_reference_text(name):
    c, db, qs, b6, uc = G.load(name)
    p, gdb, hits, nh, pool = _run_gpu(c, db, qs)
    qlens = np.diff(qs.offs.astype(np.int64))
    gb6, guc = orc.format_outputs(capi.lib(), "ugs", hits, nh, pool, qs.labels(), qlens, db.labels(), not c["aa"])
    assert gb6 == b6
    assert guc == uc


@pytest.mark.parametrize("name", ["nt_big", "hard_big", "hard_small", "hard_aa", "hard_acc"])
def test_gpu_hits_equal_oracle_records(name):
    c, db, qs, b6, uc = G.load(name)
    p, gdb, hits, nh, pool = _run_gpu(c, db, qs)
    op = orc.params(is_nucleo=not c["aa"], id=c["id"], **G.params_kw(c))
    odb = orc.OrcDB(op, db.seqs, db.offs)
    ohits, onh, opool = odb.search(qs.seqs, qs.offs, nthreads=4)
    assert np.array_equal(nh, onh)
    for f in hits.dtype.names:
        if f != "cigar_off":          # pool layout is an implementation detail; runs compared below
            assert np.array_equal(hits[f], ohits[f]), f
    for h, o in zip(hits, ohits):
        assert np.array_equal(pool[int(h["cigar_off"]):int(h["cigar_off"]) + int(h["cigar_len"])],
                              opool[int(o["cigar_off"]):int(o["cigar_off"]) + int(o["cigar_len"])])


@pytest.mark.parametrize("name", ["nt_big", "hard_big", "hard_id90", "hard_small", "nt_small", "hard_aa", "aa_small"])
def test_gpu_candidate_order_equals_oracle(name):
    c, db, qs, b6, uc = G.load(name)
    p = capi.params(is_nucleo=not c["aa"], id=c["id"], **G.params_kw(c))
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
    bat.upload(qs.seqs, qs.offs)
    bat.search(); bat.sync()
    cand, cnt, n = bat.candidates()
    op = orc.params(is_nucleo=not c["aa"], id=c["id"], **G.params_kw(c))
    odb = orc.OrcDB(op, db.seqs, db.offs)
    K = cand.shape[1]
    ns = 2 if p.strand_both else 1
    for qi in range(0, qs.n, 7):
        q = qs.seqs[int(qs.offs[qi]):int(qs.offs[qi + 1])]
        for s in range(ns):
            qq = q if s == 0 else orc.revcomp(q)
            on, ocand, ocnt = odb.rank(qq, cap=K)
            u = qi * ns + s
            m = min(on, K)
            assert n[u] == m, (qi, s, n[u], on)
            assert np.array_equal(cand[u, :m], ocand[:m]), (qi, s)
            assert np.array_equal(cnt[u, :m], ocnt[:m]), (qi, s)


def test_gpu_mask_and_index_equal_oracle():
    c, db, qs, b6, uc = G.load("hard_small")
    p = capi.params(is_nucleo=True, id=c["id"])
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    masked, row_off, postings = gdb.debug_fetch()
    odb = orc.OrcDB(orc.params(is_nucleo=True, id=c["id"]), db.seqs, db.offs)
    assert np.array_equal(masked, odb.masked())
    oro, opo = odb.index()
    assert np.array_equal(row_off, oro)
    assert np.array_equal(postings, opo)


def test_gpu_small_partitions(monkeypatch):
    """Force tiny target partitions (many per DB) so the multi-partition scan, the per-partition
    count-1 fill quota and cross-partition tie order are all exercised on a small DB."""
    monkeypatch.setenv("UGS_GSHIFT", "7")
    for name in ("hard_big", "nt_small", "hard_both"):
        c, db, qs, b6, uc = G.load(name)
        p, gdb, hits, nh, pool = _run_gpu(c, db, qs)
        qlens = np.diff(qs.offs.astype(np.int64))
        gb6, guc = orc.format_outputs(capi.lib(), "ugs", hits, nh, pool, qs.labels(), qlens, db.labels(), True)
        assert gb6 == b6, name
        assert guc == uc, name


def test_gpu_device_results_view_matches_fetch():
    """The device-resident hit table exposed for the RCCL gather (ugs_batch_device_results) holds the
    same records ugs_batch_fetch returns; it is wrapped zero-copy through __cuda_array_interface__."""
    import torch
    from usearch12_amd import multigpu
    from usearch12_amd.abi import HIT_DTYPE
    import bench
    c, db, qs, b6, uc = G.load("hard_acc")
    p = capi.params(is_nucleo=True, id=c["id"], **G.params_kw(c))
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=0)
    bat = capi.UgsBatch(gdb, qs.n, int(qs.offs[-1]))
    bat.upload(qs.seqs, qs.offs); bat.search(); bat.sync()
    hits, nh, pool = bat.fetch()
    (ph, bh), (pn, bn), (pc, bc) = bat.device_results(query_base=1000)
    t_h = torch.as_tensor(bench.DevArray(ph, bh), device="cuda").cpu().numpy()
    t_n = torch.as_tensor(bench.DevArray(pn, bn), device="cuda").cpu().numpy()
    t_c = torch.as_tensor(bench.DevArray(pc, bc), device="cuda").cpu().numpy()
    ghits, gcnt, gpool = multigpu.merge_tables([t_h], [t_n], [t_c])
    assert len(ghits) == len(hits) and np.array_equal(gcnt, nh)
    ghits["query"] -= 1000
    # fetch() re-sorts the hits of a query by score; compare as per-query sets
    key = lambda a: sorted(zip(a["query"].tolist(), a["target"].tolist(), a["ids"].tolist(), a["aln_len"].tolist()))
    assert key(ghits) == key(hits)


def test_cli_text_identical_to_reference(tmp_path):
    """The C++ driver (usearch12_amd/ugs_cli, the reference's command line for this one command):
    FASTA in, -blast6out / -uc out, byte-identical to the reference's files."""
    import subprocess
    cli = os.path.join(os.path.dirname(capi.LIB_PATH), "ugs_cli")
    for name in ("hard_both", "hard_aa"):
        c, db, qs, b6, uc = G.load(name)
        dbfa, qfa = str(tmp_path / "db.fa"), str(tmp_path / "q.fa")
        db.write_fasta(dbfa); qs.write_fasta(qfa)
        cmd = [cli, "-usearch_global", qfa, "-db", dbfa, "-id", str(c["id"]), "-blast6out", str(tmp_path / "o.b6"),
               "-uc", str(tmp_path / "o.uc"), "-batch", "500"]
        if not c["aa"]:
            cmd += ["-strand", c["strand"]]
        for opt in ("big", "maxaccepts", "maxrejects"):
            if opt in c:
                cmd += ["-" + opt, str(c[opt])]
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        assert open(tmp_path / "o.b6").read() == b6, name
        assert open(tmp_path / "o.uc").read() == uc, name
