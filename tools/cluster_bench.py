"""cluster_fast (BASELINE config C3 shape) on one GPU: reads/s, batch statistics, and a parity check of a prefix
against the oracle's serial loop (the checker; never timed as the product).
  python tools/cluster_bench.py --reads 1000000 [--species N] [--check 20000] [--ref] [--out gpurun_out/x.json]"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from usearch12_amd import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=200000)
    ap.add_argument("--species", type=int, default=0)
    ap.add_argument("--length", type=int, default=300)
    ap.add_argument("--id", type=float, default=0.97)
    ap.add_argument("--both", action="store_true")
    ap.add_argument("--check", type=int, default=0, help="also cluster the first N reads with the oracle and compare")
    ap.add_argument("--ref", action="store_true", help="time the reference binary (-threads 1) on the same reads")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    t0 = time.time()
    r = synth.make_reads(3, a.reads, n_species=a.species or None, length=a.length)
    gen_s = time.time() - t0
    p = capi.cluster_params(a.id, strand_both=a.both)
    capi.UgsCluster(p, r.slice(0, min(2000, r.n)).seqs, r.slice(0, min(2000, r.n)).offs).close()     # warm-up (module load)
    t0 = time.time()
    res = capi.UgsCluster(p, r.seqs, r.offs)
    dt = time.time() - t0
    st = res.stats
    out = dict(metric="reads/s cluster_fast -id %.2f" % a.id, value=a.reads / dt, unit="reads/s", reads=a.reads, seconds=dt,
               n_unique=res.n_unique, n_clusters=res.n_clusters, batches=st.batches, batches_cut=st.batches_cut, max_batch=st.max_batch,
               queries_redone=st.queries_redone, inbatch_entries=st.inbatch_entries, pairs_in_batch=st.pairs_in_batch,
               hits_in_batch=st.hits_in_batch, pairs_frozen=st.pairs_frozen, postings=st.postings, ms_rank=st.ms_rank,
               ms_align=st.ms_align, gen_s=gen_s, host_s={k: round(getattr(st, k), 3) for k in ("s_derep", "s_search", "s_inbatch", "s_d2h", "s_replay", "s_pairs", "s_append", "s_total")}, strand="both" if a.both else "plus")
    # the ranking kernels over the whole run against the HBM roofline: 4 bytes per posting scanned / their summed device time
    out["roofline_k_rank"] = {"bound": "hbm", "algorithmic_bytes": 4 * int(st.postings), "kernel_ms": st.ms_rank,
                              "achieved_GBps": 4 * st.postings / max(st.ms_rank * 1e-3, 1e-9) / 1e9, "peak_GBps": 8000.0,
                              "frac": 4 * st.postings / max(st.ms_rank * 1e-3, 1e-9) / 8e12}
    # size-independent properties: every unique in exactly one cluster, sizes add up, centroids are their own cluster's founder
    assert int(res.cluster_size.sum()) == a.reads
    assert np.array_equal(res.uniq_cluster[res.centroid_uniq], np.arange(res.n_clusters, dtype=np.uint32))
    assert np.all(res.uniq_nhits[res.centroid_uniq] == 0)
    assert np.all((res.uniq_nhits == 0) == np.isin(np.arange(res.n_unique), res.centroid_uniq))
    if len(res.hits):
        ident = res.hits["ids"] / np.maximum(res.hits["aln_len"], 1)
        assert ident.min() >= np.float32(a.id) - 1e-9
        assert np.all(res.hits["target"] < res.n_clusters)
        # a member's centroid was founded before it
        assert np.all(res.centroid_uniq[res.hits["target"]] < res.hits["query"])
    if a.check:
        import orc
        n = min(a.check, a.reads)
        s = r.slice(0, n)
        t0 = time.time()
        o = orc.cluster_fast(orc.cluster_params(a.id, strand_both=a.both), s.seqs, s.offs)
        out["oracle_seconds_prefix"] = time.time() - t0
        g = capi.UgsCluster(p, s.seqs, s.offs)
        ok = g.n_clusters == o.n_clusters and np.array_equal(g.uniq_cluster, o.uniq_cluster) and np.array_equal(g.hits["ids"], o.hits["ids"]) \
            and np.array_equal(g.hits["target"], o.hits["target"]) and np.array_equal(g.centroid_uniq, o.centroid_uniq)
        out["prefix_checked"] = n
        out["prefix_equal_oracle"] = bool(ok)
        assert ok
    if a.ref:
        ref = os.path.join(ROOT, "oracle", "_ref", "usearch12")
        fa = "/tmp/cluster_bench_reads.fa"
        r.write_fasta(fa)
        t0 = time.time()
        subprocess.check_call([ref, "-cluster_fast", fa, "-id", str(a.id), "-uc", "/tmp/cluster_bench.uc", "-threads", "1",
                               "-strand", "both" if a.both else "plus"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out["reference_seconds"] = time.time() - t0
        lab = r.labels()
        res.write_uc(lab, "/tmp/cluster_bench_gpu.uc")
        out["uc_identical_to_reference"] = open("/tmp/cluster_bench.uc", "rb").read() == open("/tmp/cluster_bench_gpu.uc", "rb").read()
        assert out["uc_identical_to_reference"]
    line = json.dumps(out)
    print(line)
    if a.out:
        with open(a.out, "a") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
