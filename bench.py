#!/usr/bin/env python3
"""bench.py - usearch_global hot path on MI355X (BASELINE.json metric: query-seqs/s).

One "step" = one pass of the hot path over one batch of synthetic queries: H2D upload of the batch, ranking +
alignment kernels, hit table back on the host.  The upload of step i+1's batch is issued while step i's kernels
run (two batch objects in flight, a copy stream per batch), and every step searches a batch that differs from the one
before it; both are inside the timed region.

  every N  BASELINE.json configs[1] (C2), the configuration the metric is quoted on: every GPU searches its own batch of
         1M x 250 nt queries per step (N x 1M queries per step over the job: WEAK scaling) vs the 1M-sequence DB whose
         index is replicated in every GPU's HBM, -id 0.97, -strand plus, reference defaults.  For N > 1 the query stream
         is sharded one GPU per shard with no collective on the data path; the only exchange is one gather of the
         device-resident hit tables to rank 0 per step - the product's own C++ gather (include/ugs_comm.h,
         libugs_rccl.so: ncclAllGather of the sizes + grouped ncclSend/ncclRecv over xGMI) - issued for step i while the
         kernels of step i+1 run.  The same workload at every N keeps the driver's per-N values comparable.
  --workload C4  BASELINE.json configs[3]: 10M x 250 nt queries split into N contiguous shards vs a 5M-sequence DB
         (STRONG scaling; a 5M-sequence index makes every query read 5x the postings of C2, so its lines compare with
         `--gpus 1 --workload C4`, not with the C2 lines).
`python bench.py --gpus N` starts its N ranks itself (one process per GPU, 127.0.0.1) when it is not already running under a launcher
(the driver's `python -m torch.distributed.run ... bench.py --gpus N`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), and
refuses to run when the launcher's world size differs from --gpus.  The RANK PROCESSES never import torch: torch's wheel bundles a HIP
runtime and an RCCL of its own under the system libraries' SONAMEs (usearch12_amd/hostgroup.py), so a rank rendezvouses over plain
sockets, runs ONE HIP runtime and ONE RCCL - the system ones libugs.so / libugs_rccl.so link - and reports the libraries it has mapped
(detail.runtime_libs); the barriers of the timed region are the socket barrier + ugs_device_synchronize (= hipDeviceSynchronize).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def runtime_libs():
    """the HIP / HSA / RCCL libraries this process has mapped (from /proc/self/maps) and whether torch was imported: a rank must hold
    exactly one of each - two copies of a runtime in one process corrupt each other's heap (usearch12_amd/hostgroup.py)"""
    import re
    found = {}
    try:
        for ln in open("/proc/self/maps"):
            m = re.search(r"(/\S*/(libamdhip64|libhsa-runtime64|librccl)[^/\s]*)$", ln.strip())
            if m:
                found.setdefault(m.group(2), set()).add(os.path.realpath(m.group(1)))
    except OSError:
        pass
    out = {k: sorted(v) for k, v in found.items()}
    out["torch_imported"] = "torch" in sys.modules
    return out


def blast6_lines(capi, hits, qlabel, tlabel):
    """-blast6out text of a hit table through the product's own formatter (ugs_format_blast6 = blast6out.cpp:27-80), one bytes line per hit"""
    L = capi.lib()
    buf = ctypes.create_string_buffer(1024)
    out = []
    rec = hits.dtype.itemsize
    base = hits.ctypes.data
    for i in range(len(hits)):
        n = L.ugs_format_blast6(base + i * rec, qlabel(int(hits["query"][i])).encode(), tlabel(int(hits["target"][i])).encode(), buf, 1024)
        out.append(buf.raw[:n])
    return out


def cpu_baseline(kind, db, qs, ident, sweep_q, parity_q, nproc):
    """The CPU path on bounded samples of the same workload on this host's cores (checker / baseline leg: nothing here is inside the
    timed region or on the product path).  kind "reference" = the unmodified usearch12 binary (oracle/_ref):
      * the database is indexed ONCE (-makeudb_usearch), so that a search run is file load + search, and the load is what a 1-query run takes;
      * a thread sweep (-threads 16 / 32 / 64 / 128 / all) over the first `sweep_q` queries: the reference's shared FASTA reader and output
        lock put its best point far below "all threads"  - `value` is the best point of the sweep;
      * one run over the first `parity_q` queries at the best thread count whose -blast6out is KEPT: main() compares it with the GPU's hits.
    Returns (cpu_baseline object, reference blast6 lines or None, queries of that run)."""
    if kind == "none":
        return None, None, 0
    if kind == "reference":
        ref = os.path.join(ROOT, "oracle", "_ref", "usearch12")
        if not os.path.exists(ref):
            kind = "port"
    if kind == "reference":
        with tempfile.TemporaryDirectory() as tmp:
            dbfa, q1 = os.path.join(tmp, "db.fa"), os.path.join(tmp, "q1.fa")
            db.write_fasta(dbfa)
            qs.slice(0, 1).write_fasta(q1)
            t0 = time.time()
            udb = os.path.join(tmp, "db.udb")
            rc = subprocess.call([ref, "-makeudb_usearch", dbfa, "-output", udb], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            t_makeudb = time.time() - t0
            dbarg = udb if rc == 0 and os.path.exists(udb) else dbfa

            def run(q, threads, out):
                t0 = time.time()
                subprocess.check_call([ref, "-usearch_global", q, "-db", dbarg, "-id", str(ident), "-strand", "plus",
                                       "-blast6out", out, "-threads", str(threads)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                return time.time() - t0
            sweep_n = min(sweep_q, qs.n)
            qsw = os.path.join(tmp, "qsweep.fa")
            qs.slice(0, sweep_n).write_fasta(qsw)
            points = []
            for T in sorted({t for t in (16, 32, 64, 128, nproc) if 1 <= t <= nproc}):
                t_load = run(q1, T, os.path.join(tmp, "o1.b6"))
                t_full = run(qsw, T, os.path.join(tmp, "osw.b6"))
                points.append({"threads": T, "queries": sweep_n, "run_s": round(t_full, 3), "load_s": round(t_load, 3),
                               "query_seqs_per_s": sweep_n / max(t_full - t_load, 1e-3)})
            best = max(points, key=lambda x: x["query_seqs_per_s"])
            par_n = min(parity_q, qs.n)
            lines = None
            if par_n:
                qpar = os.path.join(tmp, "qpar.fa")
                qs.slice(0, par_n).write_fasta(qpar)
                t_load = run(q1, best["threads"], os.path.join(tmp, "o1.b6"))
                t_full = run(qpar, best["threads"], os.path.join(tmp, "opar.b6"))
                points.append({"threads": best["threads"], "queries": par_n, "run_s": round(t_full, 3), "load_s": round(t_load, 3),
                               "query_seqs_per_s": par_n / max(t_full - t_load, 1e-3), "blast6out_kept_for_parity_sample": True})
                lines = open(os.path.join(tmp, "opar.b6"), "rb").read().splitlines(keepends=True)
                best = max(points, key=lambda x: x["query_seqs_per_s"])
        return ({"value": best["query_seqs_per_s"], "unit": "query-seqs/s", "cores": best["threads"], "kind": "reference",
                 "host_threads_available": nproc, "sweep": points,
                 "sample": "best point of a -threads sweep of the unmodified usearch12 binary (oracle/build_ref.sh: g++ -O3 -march=x86-64-v2 on the "
                           "reference's own sources, not its Makefile's -march=native) over the first %d of the same C2 queries vs the full %d-seq DB "
                           "(indexed once with -makeudb_usearch, %.1f s); search wall of a point = its run minus a 1-query run at the same thread "
                           "count (file load); %d queries at the best thread count for the parity sample" % (sweep_n, db.n, t_makeudb, par_n)},
                lines, par_n)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc   # the CPU restatement: checker / baseline only, never the product path
    sample = qs.slice(0, min(sweep_q, qs.n))
    p = orc.params(is_nucleo=True, id=ident)
    odb = orc.OrcDB(p, db.seqs, db.offs)
    t0 = time.time()
    odb.search(sample.seqs, sample.offs, nthreads=nproc)
    t = time.time() - t0
    return ({"value": sample.n / t, "unit": "query-seqs/s", "cores": nproc, "kind": "port",
             "sample": "%d of the same C2 queries vs the full %d-seq DB, oracle/ugs_oracle.c with %d threads" % (sample.n, db.n, nproc)}, None, 0)


REF_BIN = os.path.join(ROOT, "oracle", "_ref", "usearch12")


def ref_parity_global(capi, db, qs, ident, gpu_hits, par_n, threads=16, what=""):
    """checker leg (never inside a timed region): the unmodified reference binary's own -blast6out for the first `par_n` queries of `qs`
    against ugs_format_blast6 of the GPU hit table `gpu_hits` for the same queries, as sorted multisets of lines (blast6out.cpp:27-80)"""
    if not os.path.exists(REF_BIN):
        return {"skipped": "oracle/_ref/usearch12 is not on this box"}
    t0 = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        dbfa, qfa, udb, out = (os.path.join(tmp, x) for x in ("db.fa", "q.fa", "db.udb", "o.b6"))
        db.write_fasta(dbfa)
        qs.slice(0, par_n).write_fasta(qfa)
        rc = subprocess.call([REF_BIN, "-makeudb_usearch", dbfa, "-output", udb], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dbarg = udb if rc == 0 and os.path.exists(udb) else dbfa
        t1 = time.time()
        subprocess.check_call([REF_BIN, "-usearch_global", qfa, "-db", dbarg, "-id", str(ident), "-blast6out", out, "-threads", str(threads)],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t2 = time.time()
        theirs = sorted(open(out, "rb").read().splitlines(keepends=True))
    mine = sorted(blast6_lines(capi, gpu_hits, qs.label, db.label))
    same = mine == theirs
    res = {"queries": par_n, "hits_gpu": len(mine), "hits_reference": len(theirs), "identical": bool(same), "reference_threads": threads,
           "reference_search_s": round(t2 - t1, 2), "leg_s": round(time.time() - t0, 1),
           "what": what + "every -blast6out line of the unmodified reference binary for the first %d queries vs ugs_format_blast6 of the GPU hit table "
                          "for the same queries, as sorted multisets of lines" % par_n}
    if not same:
        res["first_differences"] = [x.decode(errors="replace") for x in sorted(set(mine) ^ set(theirs))[:4]]
    return res


def ref_parity_cluster(capi, reads, ident, par_n, device):
    """checker leg: `usearch12 -cluster_fast -id .. -uc -threads 1` (the reference's deterministic setting) on the first `par_n` reads against
    ugs_cluster_write_uc of the GPU run on the same prefix, byte for byte (outputuc.cpp:45-93, clusterfast.cpp:81-133)"""
    if not os.path.exists(REF_BIN):
        return {"skipped": "oracle/_ref/usearch12 is not on this box"}
    t0 = time.time()
    sub = reads.slice(0, par_n)
    with tempfile.TemporaryDirectory() as tmp:
        fa, ruc, guc = (os.path.join(tmp, x) for x in ("r.fa", "ref.uc", "gpu.uc"))
        sub.write_fasta(fa)
        t1 = time.time()
        subprocess.check_call([REF_BIN, "-cluster_fast", fa, "-id", str(ident), "-uc", ruc, "-threads", "1", "-strand", "plus"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t2 = time.time()
        res = capi.UgsCluster(capi.cluster_params(ident), sub.seqs, sub.offs, device=device)
        res.write_uc(sub.labels(), guc)
        ncl = int(res.n_clusters)
        res.close()
        a, b = open(ruc, "rb").read(), open(guc, "rb").read()
    out = {"reads": par_n, "identical": bool(a == b), "uc_bytes": len(a), "clusters": ncl, "reference_s_1_thread": round(t2 - t1, 2), "leg_s": round(time.time() - t0, 1),
           "what": "C3: the -uc file of the unmodified reference binary (-cluster_fast -id %s -threads 1) for the first %d reads vs ugs_cluster_write_uc of the "
                   "GPU run on the same reads, byte for byte" % (ident, par_n)}
    if a != b:
        la, lb = a.splitlines(), b.splitlines()
        k = next((i for i, (x, y) in enumerate(zip(la, lb)) if x != y), min(len(la), len(lb)))
        out["first_difference"] = {"line": k, "reference": la[k].decode(errors="replace") if k < len(la) else None, "gpu": lb[k].decode(errors="replace") if k < len(lb) else None}
    return out


def c4_on_one_device(capi, synth, device, world=8, total_q=10_000_000, db_n=5_000_000, length=250):
    """BASELINE.json configs[3] on the ONE GPU the driver's bench box has: the 10 M-query stream in `world` contiguous shards, one rank
    (batch object + communicator rank + host thread for the collective) per shard, the shards searched one after the other against the
    5 M-sequence index, then ONE gather of the eight device-resident hit tables to rank 0 (ugs_gather_results, loopback transport:
    device-to-device copies where the 8-GPU job has ncclSend / ncclRecv).  value = 10 M / (first upload -> merged table on the host)."""
    import threading
    from usearch12_amd import multigpu
    t0 = time.time()
    db = synth.make_db(4, db_n, length)
    p = capi.params(is_nucleo=True, id=0.97)
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=device)
    bounds = [multigpu.shard_range(total_q, world, r) for r in range(world)]
    shards = [make_query_sets(synth, 4, db, lo, hi - lo, length, 1)[0] for lo, hi in bounds]
    gen_s = time.time() - t0
    bats = []
    for q in shards:
        capi._chk(capi.lib().ugs_host_register(q.seqs.ctypes.data, q.seqs.nbytes))
        bats.append(capi.UgsBatch(gdb, q.n, int(q.offs[-1])))
    comms = capi.UgsComm.init_loopback(world, device)
    best = None
    for rep in range(2):                                          # (the first pass sizes every scratch buffer)
        res, err = [None] * world, []
        t1 = time.time()
        for b, q in zip(bats, shards):
            b.upload(q.seqs, q.offs); b.search()
        for b in bats:
            b.sync()
        t2 = time.time()

        def run(r):
            try:
                res[r] = comms[r].gather(bats[r], bounds[r][0], dst=0)
            except Exception as e:
                err.append("%d: %s" % (r, e))
        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        [t.start() for t in th]; [t.join(600) for t in th]
        t3 = time.time()
        if err:
            raise RuntimeError("; ".join(err))
        best = (t3 - t1, t2 - t1, t3 - t2)
    hits, nh, pool = res[0]
    sts = [b.stats() for b in bats]
    khs = [b.kernel_hits() for b in bats]
    ms_rank = sum(x["ms_rank"] for x in sts)
    b_rank = sum(4 * x["postings"] + x["query_letters"] for x in sts)
    ok = len(nh) == total_q and int(nh.sum()) == len(hits) and bool(np.all(np.diff(hits["query"].astype(np.int64)) > 0))
    out = {"workload": "C4: usearch_global %d x %d nt queries in %d contiguous shards (one rank each, all on this one GPU, searched one after the other) vs the "
                       "%d-seq DB, -id 0.97, one gather of the %d hit tables to rank 0 through ugs_gather_results (loopback transport)" % (total_q, length, world, db_n, world),
           "value": total_q / best[0], "unit": "query-seqs/s", "seconds": best[0], "seconds_search": best[1], "seconds_gather_to_host": best[2],
           "hits": int(len(hits)), "merged_table_consistent": bool(ok),
           "kernel_ms": {"ranking (8 shards)": ms_rank, "k_align (8 shards)": sum(x["ms_align"] for x in sts), "k_rank_setup (8 shards)": sum(x["ms_rank_setup"] for x in sts)},
           "kernel": "k_rank2 + k_rank over %d deferred units" % sum(k["deferred"] for k in khs), "algorithmic_bytes": int(b_rank),
           "frac": b_rank / (ms_rank * 1e-3) / (HBM_PEAK_GBS * 1e9), "predicted_8_gpu_value": total_q / (best[1] / world + best[2]), "gen_s": gen_s}
    for q in shards:
        capi.lib().ugs_host_unregister(q.seqs.ctypes.data)
    for b in bats:
        b.close()
    for c in comms:
        c.close()
    gdb.close()
    return out


def other_config(capi, synth, name, device, parity=True):
    """One of BASELINE.json's other configurations, once, on this GPU (driver-visible numbers for what DESIGN.md section 4 quotes; VERDICT
    r04 item 2).  Same step as the bench line: upload + kernels + fetch; 1 warm-up + 3 timed steps of one batch."""
    t0 = time.time()
    if name == "C4":
        return c4_on_one_device(capi, synth, device)
    if name == "C3":
        r = synth.make_reads(3, 5_000_000, length=300)
        gen_s = time.time() - t0
        p = capi.cluster_params(0.97)
        capi.UgsCluster(p, r.slice(0, 2000).seqs, r.slice(0, 2000).offs).close()           # (module load)
        t0 = time.time()
        res = capi.UgsCluster(p, r.seqs, r.offs, device=device)
        dt = time.time() - t0
        st = res.stats
        out = {"workload": "C3: cluster_fast 5000000 x 300 nt reads -id 0.97 (UCLUST centroid path), one MI355X", "value": r.n / dt, "unit": "reads/s",
               "seconds": dt, "clusters": int(res.n_clusters), "uniques": int(res.n_unique),
               "kernel_ms": {"ranking (all batches)": st.ms_rank, "k_align (all batches)": st.ms_align},
               "kernel": "k_rank2<cluster_fast>, k_rank2<HV> over the units that defers (%d of them ranked there), k_rank behind both and on the small path" % int(st.units_heavy),
               "units_ranked_by_the_heavy_unit_kernel": int(st.units_heavy), "algorithmic_bytes": 4 * int(st.postings),
               "frac": 4 * st.postings / max(st.ms_rank * 1e-3, 1e-9) / (HBM_PEAK_GBS * 1e9), "gen_s": gen_s}
        res.close()
        if parity:
            out["parity_sample"] = ref_parity_cluster(capi, r, 0.97, 150_000, device)
        return out
    if name == "ID90":
        # the C2 shape at -id 0.9 (not a BASELINE configuration): ~ 41 sampled index rows per query instead of ~ 11 - the
        # general ranking kernel's mid-identity instantiation (8-bit counters), outside the bitmap kernel's 15 rows
        db = synth.make_db(2, 1_000_000, 250); qs = synth.make_queries(2, db, 1_000_000, 250)
        p = capi.params(is_nucleo=True, id=0.9)
        wl = "ID90: usearch_global 1000000 x 250 nt queries vs 1000000-seq DB, -id 0.9 (mid-identity: ~41 sampled rows per query), one MI355X"
        parity = False
    elif name == "C5":
        db = synth.make_db(5, 2_000_000, 300, aa=True); qs = synth.make_queries(5, db, 1_000_000, 300, aa=True)
        p = capi.params(is_nucleo=False, id=0.8)
        wl = "C5: usearch_global protein 1000000 x 300 aa queries vs 2000000-seq aa DB, -id 0.8, one MI355X"
    else:
        raise ValueError("unknown configuration %r" % name)
    gen_s = time.time() - t0
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=device)
    bats = [capi.UgsBatch(gdb, qs.n, int(qs.offs[-1])) for _ in range(2)]
    capi._chk(capi.lib().ugs_host_register(qs.seqs.ctypes.data, qs.seqs.nbytes))
    # the bench line's own pipeline: two batch objects, step i + 1 enqueued before the host waits for step i, every step uploads its batch
    nsteps, st, kh, nh = 6, None, None, 0
    for b in bats:
        b.upload(qs.seqs, qs.offs)
    bats[0].search(); bats[0].sync(); bats[0].fetch(reuse=True)           # (sizes the scratch buffers)
    bats[0].upload(qs.seqs, qs.offs); bats[0].sync_upload(); bats[1].sync_upload()
    t1 = time.time()
    bats[0].search()
    for i in range(nsteps):
        cur, nxt = bats[i % 2], bats[(i + 1) % 2]
        if i + 1 < nsteps:
            nxt.search()
        cur.sync()
        hits = cur.fetch(reuse=True)[0]
        st, kh, nh = cur.stats(), cur.kernel_hits(), len(hits)
        if i + 2 < nsteps:
            cur.upload(qs.seqs, qs.offs)
    dt = (time.time() - t1) / nsteps
    capi.lib().ugs_host_unregister(qs.seqs.ctypes.data)
    b_rank = 4 * st["postings"] + st["query_letters"]
    kern = kh.get("r2_kernel") or "k_rank"
    out = {"workload": wl, "value": qs.n / dt, "unit": "query-seqs/s", "ms_per_step": 1000 * dt, "hits_per_step": int(nh),
           "kernel_ms": {"ranking": st["ms_rank"], "k_align": st["ms_align"], "k_rank_setup": st["ms_rank_setup"]},
           "kernel": kern + " + k_rank over %d deferred units" % kh["deferred"], "algorithmic_bytes": int(b_rank),
           "frac": b_rank / (st["ms_rank"] * 1e-3) / (HBM_PEAK_GBS * 1e9), "gen_s": gen_s}
    if parity:
        par_n = 100_000
        sub = qs.slice(0, par_n)
        bats[0].upload(sub.seqs, sub.offs); bats[0].search(); bats[0].sync()
        h, _nh, _pool = bats[0].fetch()
        out["parity_sample"] = ref_parity_global(capi, db, qs, 0.8, h, par_n, what="C5: ")
    for b in bats:
        b.close()
    gdb.close()
    return out


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


QCHUNK = 250_000


def make_query_sets(synth, seed, db, lo, n, length, nsets):
    """`nsets` different batches: queries [lo, lo + n) of a GLOBAL query stream per set.  The stream is generated in chunks of
    QCHUNK queries seeded by (set, chunk index), so any sharding of it yields the same queries - a 2-rank run and a 1-rank run
    of the same total search the same set (what the multi-rank test checks)."""
    import numpy as _np
    sets = []
    for k in range(nsets):
        seqs, lens = [], []
        pos = lo
        while pos < lo + n:
            c = pos // QCHUNK
            q = synth.make_queries(seed + 7919 * k + 104729 * c, db, QCHUNK, length)
            a, b = pos - c * QCHUNK, min(lo + n, (c + 1) * QCHUNK) - c * QCHUNK
            seqs.append(q.seqs[int(q.offs[a]):int(q.offs[b])])
            lens.append(_np.diff(q.offs[a:b + 1].astype(_np.int64)))
            pos = c * QCHUNK + b
        ln = _np.concatenate(lens) if lens else _np.zeros(0, _np.int64)
        offs = _np.zeros(len(ln) + 1, _np.uint64)
        offs[1:] = _np.cumsum(ln).astype(_np.uint64)
        sets.append(synth.SeqSet(_np.ascontiguousarray(_np.concatenate(seqs)) if seqs else _np.zeros(0, _np.uint8), offs, lambda i: "q%d" % i))
    return sets


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["auto", "C2", "C4"], default="auto", help="auto = C2 (weak scaling: 1M queries per GPU and step); "
                    "C4 = 10M queries in N shards vs a 5M-sequence DB (strong scaling)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="C2 only: weak = every GPU its own batch of the C2 size per step "
                    "(the default, what the driver's per-N lines compare); strong = the SAME batch of the C2 size split into N contiguous "
                    "shards (125 k queries per GPU at N = 8: set-up, launch gaps and the gather are then a visible share of a step)")
    ap.add_argument("--force-gather", action="store_true", help="--gpus 1 only: run the N > 1 step (C++ gather through a communicator "
                    "of one rank) instead of the plain fetch - exercises that code on a one-GPU box")
    ap.add_argument("--batches", type=int, default=2, help="batch objects of the pipeline (searches enqueued ahead = batches - 1).  Measured in r6 with "
                    "three (the host's gather of step i then never delays the enqueue of a later step): C2 48.5 instead of 45.9 ms per step, the strong-"
                    "scaling proxy 8.1 instead of 8.2 ms - the proxy's 1.3 ms outside the search kernels is GPU-side work of the emulated peers, not host latency")
    ap.add_argument("--db", type=int, default=0, help="DB sequences (default: the workload's)")
    ap.add_argument("--queries", type=int, default=0, help="queries per step: per GPU for C2 (weak), over all GPUs for C4 (default: the workload's)")
    ap.add_argument("--length", type=int, default=250)
    ap.add_argument("--id", type=float, default=0.97)
    ap.add_argument("--cpu-baseline", choices=["reference", "port", "none"], default="reference")
    ap.add_argument("--cpu-sample", type=int, default=48_000, help="queries of every point of the CPU thread sweep")
    ap.add_argument("--parity-sample", type=int, default=384_000, help="queries the reference binary searches with its -blast6out kept: the GPU's "
                    "hits for the same queries are formatted and compared with it (0 = off)")
    ap.add_argument("--other-configs", default="auto", help="comma list of C5,C4,C3,ID90 run once each after the timed region (detail.other_configs); "
                    "auto = all three on a default one-GPU C2 run, none otherwise")
    ap.add_argument("--emulate-world", type=int, default=8, help="one-GPU C2 runs: strong-scaling proxy - rank 0's step at 1/N of the batch with the "
                    "N-rank gather (loopback transport), detail.strong_scaling_proxy; 0 = off")
    ap.add_argument("--backend", choices=["nccl", "host", "gloo"], default="nccl",
                    help="nccl: the product's gather over RCCL (libugs_rccl.so), the measurement.  host (old name: gloo): functional dry run of "
                         "the N > 1 path on a box with fewer GPUs than ranks (ranks share GPUs, tables travel through the host) - not a measurement")
    args = ap.parse_args()

    # ONE line on stdout: whatever the libraries below print there (RCCL's version banner at communicator creation, for one) goes to
    # stderr; the JSON line is written to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start the N ranks ourselves (one process per GPU; RCCL over xGMI between them)
        port = free_port()
        procs = []
        for r in range(args.gpus):
            env = dict(os.environ)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=real_stdout))
        rc = 0
        while procs:                                                # a rank that fails takes the others with it (they would wait in a barrier)
            time.sleep(0.05)
            for p_ in list(procs):
                r_ = p_.poll()
                if r_ is None:
                    continue
                procs.remove(p_)
                if r_ != 0 and rc == 0:
                    rc = abs(r_) or 1
                    for q_ in procs:
                        q_.terminate()
        sys.exit(rc)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # (dmabuf IPC: RCCL between processes needs it on this driver)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE); refusing to report a mislabelled line" % (args.gpus, world))
    if args.backend == "gloo":
        args.backend = "host"                                       # (the old name of the dry-run transport)

    from usearch12_amd import capi, synth, multigpu
    from usearch12_amd.build import csrc_hash
    from usearch12_amd.hostgroup import SocketGroup
    csrc_sha = csrc_hash()
    capi.lib()
    group = None
    if world > 1:
        ndev = capi.device_count()
        if args.backend == "host":
            local_rank = local_rank % max(ndev, 1)                  # dry run: the ranks share the GPUs there are
        elif local_rank >= ndev:
            sys.exit("bench.py rank %d: LOCAL_RANK %d but %d GPU(s) visible (use --backend host for a functional dry run on fewer GPUs)" % (rank, local_rank, ndev))
        group = SocketGroup(rank, world)

    workload = args.workload if args.workload != "auto" else "C2"
    seed = 2 if workload == "C2" else 4
    db_n = args.db or (1_000_000 if workload == "C2" else 5_000_000)
    strong_c2 = workload == "C2" and args.scaling == "strong"
    if strong_c2:                                                   # strong: ONE batch of the C2 size in N contiguous shards
        total_q = args.queries or 1_000_000
        lo, hi = multigpu.shard_range(total_q, world, rank)
        shard_n = hi - lo
    elif workload == "C2":                                          # weak: every rank its own batch of the C2 size
        shard_n = args.queries or 1_000_000
        total_q = shard_n * world
        lo = rank * shard_n
    else:                                                           # strong: the C4 query set in contiguous shards
        total_q = args.queries or 10_000_000
        lo, hi = multigpu.shard_range(total_q, world, rank)
        shard_n = hi - lo

    # ---- synthetic workload; the DB is the same on every rank (replicated index): rank 0 generates it once and the other
    # ranks of the node map the file.  The queries are the rank's shard; two different batches alternate from step to step.
    t0 = time.time()
    if world > 1:
        shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
        tag = os.path.join(shm, "ugs_bench_db_%s_%d_%d_%d" % (os.environ.get("MASTER_PORT", "0"), seed, db_n, args.length))
        if rank == 0:
            db0 = synth.make_db(seed, db_n, args.length)
            np.save(tag + "_seqs.npy", db0.seqs); np.save(tag + "_offs.npy", db0.offs)
            del db0
        group.barrier()
        db = synth.SeqSet(np.load(tag + "_seqs.npy", mmap_mode="r"), np.load(tag + "_offs.npy", mmap_mode="r"), lambda i: "t%d" % i)
    else:
        db = synth.make_db(seed, db_n, args.length)
    qsets = make_query_sets(synth, seed, db, lo, shard_n, args.length, 2)
    t_gen = time.time() - t0

    p = capi.params(is_nucleo=True, id=args.id)
    t0 = time.time()
    gdb = capi.UgsDB(p, db.seqs, db.offs, device=local_rank)
    t_index = time.time() - t0
    max_letters = max(int(q.offs[-1]) for q in qsets)
    bats = [capi.UgsBatch(gdb, shard_n, max_letters) for _ in range(max(2, args.batches))]
    for q in qsets:                                                 # page-locked once: uploads are then true async DMA
        capi._chk(capi.lib().ugs_host_register(q.seqs.ctypes.data, q.seqs.nbytes))
    if world > 1:
        group.barrier()                                             # every rank has built its index and its queries from the mapped file
        if rank == 0:
            os.remove(tag + "_seqs.npy"); os.remove(tag + "_offs.npy")
    # ---- the exchange.  nccl backend (the measurement): the product's C++ gather over its own RCCL communicator; the unique id
    # travels through the rank group, which is otherwise only rendezvous, barrier and max-reduction.  A rank without a communicator
    # FAILS the run (r5 fell back to another gather silently: the line would then not have measured the product's).
    # host backend (dry run on a box with fewer GPUs than ranks): every rank fetches its table, the tables travel through the host.
    comm = None
    use_cpp_gather = (world > 1 and args.backend == "nccl") or (world == 1 and args.force_gather)
    if use_cpp_gather:
        uid = capi.UgsComm.unique_id() if rank == 0 else None
        if world > 1:
            uid = group.broadcast(uid, 0)
        comm_err = None
        try:
            comm = capi.UgsComm.init_rank(uid, rank, world, local_rank)
        except Exception as e:
            comm_err = "%s: %s" % (type(e).__name__, e)
        errs = group.allgather(comm_err) if world > 1 else [comm_err]
        if any(errs):
            sys.exit("bench.py rank %d: the product's communicator (ugs_comm_init_rank, libugs_rccl.so) could not be created on every rank: %r" % (rank, errs))
    libs = runtime_libs()
    all_libs = group.allgather(libs) if world > 1 else [libs]
    for r_, l_ in enumerate(all_libs):
        if l_["torch_imported"] or any(len(l_.get(k, [])) > 1 for k in ("libamdhip64", "libhsa-runtime64", "librccl")):
            sys.exit("bench.py: rank %d maps more than one copy of a GPU runtime library (or imported torch): %r" % (r_, l_))
    if comm is not None:
        for b in bats:
            b.set_query_base(lo)                                    # the search's own grouping stamps global query ids
    gbuf = None
    if comm is not None and rank == 0:                              # rank 0's result buffers: page-locked once, reused every step
        cap_h = total_q * (p.max_accepts or 64) + 1
        from usearch12_amd.abi import HIT_DTYPE
        gbuf = [np.zeros(cap_h, HIT_DTYPE), np.zeros(total_q + 1, np.uint32), np.zeros(8 * total_q + 4096, np.uint32)]   # (runs: grown on demand)
        for a in gbuf:
            capi._chk(capi.lib().ugs_host_register(a.ctypes.data, a.nbytes))
    t0 = time.time()
    for k, b in enumerate(bats):                                    # one search per batch object: sizes its scratch buffers outside the timed region
        b.upload(qsets[k % 2].seqs, qsets[k % 2].offs)
        b.search(); b.sync()
    t_upload = time.time() - t0

    def barrier():
        """both sides of the timed region: every rank's device idle, then all ranks together"""
        capi._chk(capi.lib().ugs_device_synchronize(local_rank))
        if group is not None:
            group.barrier()
            capi._chk(capi.lib().ugs_device_synchronize(local_rank))

    t_parts = {"search_sync": 0.0, "fetch": 0.0, "gather": 0.0, "upload_issue": 0.0, "sync_wait": 0.0, "gather_exchange": 0.0, "gather_d2h": 0.0}

    def collect(b):
        """the hit table of a synced batch to the host: plain fetch (one GPU), the C++ gather to rank 0 (N ranks over RCCL),
        or the rank group through the host (dry run)"""
        if comm is not None:
            tg = time.time()
            try:
                got = comm.gather_into(b, lo, 0, *(gbuf if rank == 0 else (np.zeros(1, np.uint8),) * 3))
            except capi.UgsError as e:                              # rank 0's run pool too small: the exchange itself is done on every
                if e.code != -5 or rank != 0:                       # rank, so rank 0 alone grows its buffer and copies again
                    raise
                capi.lib().ugs_host_unregister(gbuf[2].ctypes.data)
                gbuf[2] = np.zeros(int(comm.last_demand()[2] * 1.25) + 4096, np.uint32)
                capi._chk(capi.lib().ugs_host_register(gbuf[2].ctypes.data, gbuf[2].nbytes))
                got = comm.refetch_into(*gbuf)
            t_parts["gather"] += time.time() - tg
            ex, d2h = comm.last_times()
            t_parts["gather_exchange"] += ex; t_parts["gather_d2h"] += d2h
            return got
        if group is None:
            tf = time.time()
            out = b.fetch(reuse=True)
            t_parts["fetch"] += time.time() - tf
            return out
        tg = time.time()                                             # host transport (dry run): fetch, global query ids, gather through the rank group
        h, nh, pool = b.fetch(reuse=True)
        h = h.copy()
        h["query"] += np.uint32(lo)
        got = multigpu.gather_tables(group, h, nh, pool, dst=0)
        t_parts["gather"] += time.time() - tg
        return multigpu.merge_tables(*got) if rank == 0 else None

    def step(i, n):
        """step i of a run of n: the searches of the next NB - 1 steps are already enqueued BEHIND step i's kernels when the host waits
        for step i (the GPU never waits for the host between steps - with three batch objects not even while step i's hit table
        travels: r6, a trace of the strong-scaling proxy showed the GPU idle for the host's gather + upload with two), then step i's
        hit table travels (to the host; for N > 1 first GPU to GPU to rank 0) while the next steps' kernels run, then the batch of step
        i + NB is uploaded into the batch object that has just been drained."""
        NB = len(bats)
        cur = bats[i % NB]
        ta = time.time()
        if i + NB - 1 < n:
            bats[(i + NB - 1) % NB].search()                        # enqueue: waits (on the GPU) for its upload and for the kernels in front
        tw = time.time()
        cur.sync()
        t_parts["sync_wait"] += time.time() - tw                    # the host waiting for step i's kernels
        out = collect(cur)
        st, kh = cur.stats(), cur.kernel_hits()                     # (this step's counters and event times: the upload below starts a new batch)
        tu = time.time()
        if i + NB < n:
            q = qsets[(i + NB) % 2]
            cur.upload(q.seqs, q.offs)                              # step i + NB's batch travels while the steps in between run
        t_parts["upload_issue"] += time.time() - tu
        t_parts["search_sync"] += time.time() - ta
        return out, st, kh

    def run_steps(n, timed):
        """every batch object uploaded (inputs resident in HBM), then - between barriers when timed - the first NB - 1 searches enqueued and
        EXACTLY n steps; step j searches batch object j % NB, which holds query set j % 2 (a batch different from the one before)"""
        NB = len(bats)
        for k, b in enumerate(bats):
            b.upload(qsets[k % 2].seqs, qsets[k % 2].offs)
        for b in bats:
            b.sync_upload()
        for k in t_parts:
            t_parts[k] = 0.0
        if timed:
            barrier()
        t0 = time.time()
        stats, khits, out = [], [], None
        for j in range(min(NB - 1, n)):
            bats[j].search()
        for i in range(n):
            out, st, kh = step(i, n)
            stats.append(st)
            khits.append(kh)
        if timed:
            barrier()
        return time.time() - t0, stats, khits, out

    def timed_steps(n_steps, n_warm):
        """n_warm untimed steps, then EXACTLY n_steps steps between barriers; returns (seconds, stats, kernel hits, last table)"""
        if n_warm:
            run_steps(n_warm, False)
        return run_steps(n_steps, True)

    elapsed, stats, khits, out = timed_steps(args.steps, args.warmup)
    n_hits_main = int(len(out[0])) if out is not None else 0    # (the table lives in buffers the batch objects own)
    per_rank = None
    if group is not None:
        elapsed = group.max(elapsed)                                 # the slowest rank's clock between the two barriers
        ns = max(args.steps, 1)
        mine = [float(np.mean([s["ms_rank"] for s in stats])), float(np.mean([s["ms_align"] for s in stats])),
                float(np.mean([s["ms_rank_setup"] for s in stats])), 1000.0 * t_parts["gather"] / (ns + 1),
                1000.0 * t_parts["gather_exchange"] / (ns + 1), 1000.0 * t_parts["gather_d2h"] / (ns + 1),
                1000.0 * t_parts["sync_wait"] / ns, 1000.0 * t_parts["search_sync"] / ns, float(shard_n)]
        allr = group.allgather(mine)
        # ms_gather: host time inside one gather call (exchange over xGMI + rank 0's device-to-host copies); it runs beside the next
        # step's kernels, ms_sync_wait_after = how long that step's kernels still ran when the gather had returned (0 = exposed)
        per_rank = [dict(zip(("ms_rank", "ms_align", "ms_rank_setup", "ms_gather", "ms_gather_exchange", "ms_gather_d2h",
                              "ms_sync_wait_after", "ms_step_host", "queries"), [float(x) for x in t])) for t in allr]

    steps = max(args.steps, 1)
    value = total_q * steps / elapsed
    qs = qsets[0]
    t_main = dict(t_parts)                                        # (the legs below run more steps through the same closures)
    db_hbm_bytes = gdb.stats()["hbm_bytes"]
    nproc = os.cpu_count() or 1
    one_gpu_c2 = world == 1 and workload == "C2" and not strong_c2 and not args.force_gather
    cb, parity, proxy, others = None, None, None, None
    if one_gpu_c2:
        # ---- checker leg 1: the reference binary on this box's host cores (thread sweep) and ITS OWN -blast6out for the first
        # parity_sample queries against the GPU's hits for the same queries.  Nothing of it is inside the timed region above.
        cb, ref_lines, par_n = cpu_baseline(args.cpu_baseline, db, qs, args.id, args.cpu_sample, args.parity_sample, nproc)
        if ref_lines is not None and par_n:
            sub = qs.slice(0, par_n)
            bats[0].upload(sub.seqs, sub.offs); bats[0].search(); bats[0].sync()
            h, nh, _pool = bats[0].fetch()
            mine = sorted(blast6_lines(capi, h, lambda i: "q%d" % i, lambda t: "t%d" % t))
            theirs = sorted(ref_lines)                            # (the reference's threads write in completion order: compare as multisets)
            same = mine == theirs
            parity = {"queries": par_n, "hits_gpu": len(mine), "hits_reference": len(theirs), "identical": bool(same),
                      "what": "every -blast6out line (blast6out.cpp:27-80) of the unmodified reference binary for the first %d queries of this run's "
                              "batch vs ugs_format_blast6 of the GPU hit table for the same queries, as sorted multisets of lines" % par_n}
            if not same:
                diff = sorted(set(mine) ^ set(theirs))[:4]
                parity["first_differences"] = [x.decode(errors="replace") for x in diff]
        # ---- strong-scaling proxy: what rank 0 of an N-GPU strong-scaling run does per step - search 1/N of the batch,
        # take part in the N-rank gather (loopback transport: device-to-device copies stand in for ncclSend/ncclRecv over xGMI), fetch
        # the WHOLE table - against the one-GPU step of the whole batch.  Ranks 1..N-1 are host threads that hold their searched shards.
        N = args.emulate_world
        if N and N > 1 and shard_n >= N:
            import threading
            n8 = shard_n // N
            sub_sets = [q.slice(0, n8) for q in qsets]
            full_bats, full_sets = list(bats), list(qsets)
            comms = capi.UgsComm.init_loopback(N, local_rank)
            peers = []
            for r in range(1, N):                                  # the other ranks' shards: searched once, then only gathered
                pq = qsets[0].slice(r * n8, (r + 1) * n8)
                pb = capi.UgsBatch(gdb, pq.n, int(pq.offs[-1]))
                pb.upload(pq.seqs, pq.offs); pb.search(); pb.sync()
                pb.set_query_base(r * n8)
                peers.append(pb)
            psteps, pwarm = max(args.steps, 10), 2
            small = [capi.UgsBatch(gdb, n8, max(int(q.offs[-1]) for q in sub_sets)) for _ in range(len(bats))]
            # (the shard's letters are views of the page-locked full batches)
            from usearch12_amd.abi import HIT_DTYPE
            cap_h = shard_n * (p.max_accepts or 64) + 1
            gbuf = [np.zeros(cap_h, HIT_DTYPE), np.zeros(shard_n + 1, np.uint32), np.zeros(16 * shard_n + 4096, np.uint32)]
            for a in gbuf:
                capi._chk(capi.lib().ugs_host_register(a.ctypes.data, a.nbytes))
            perr = []

            def peer(r):
                try:
                    dummy = (np.zeros(1, np.uint8),) * 3
                    for _ in range(pwarm + psteps):
                        comms[r].gather_into(peers[r - 1], r * n8, 0, *dummy)
                except Exception as e:
                    perr.append(e)
            th = [threading.Thread(target=peer, args=(r,)) for r in range(1, N)]
            [t.start() for t in th]
            bats[:] = small; qsets[:] = sub_sets; comm = comms[0]
            for b in small:
                b.set_query_base(0)
            t_small, st_s, _kh, got = timed_steps(psteps, pwarm)
            [t.join(600) for t in th]
            t_s = dict(t_parts)
            comm = None
            bats[:] = full_bats; qsets[:] = full_sets
            ms_full, ms_small = 1000.0 * elapsed / steps, 1000.0 * t_small / psteps
            kern_small = float(np.mean([x["ms_rank"] + x["ms_align"] + x["ms_rank_setup"] for x in st_s]))
            proxy = {"emulated_world": N, "queries_per_rank": n8, "steps": psteps,
                     "ms_step_whole_batch_one_gpu": ms_full, "ms_step_rank0_at_1_over_N": ms_small,
                     "predicted_speedup_%d" % N: ms_full / ms_small,
                     "ms_kernels_rank0": kern_small, "ms_fixed_cost_rank0": ms_small - kern_small,
                     "ms_gather_call": 1000.0 * t_s["gather"] / psteps, "ms_gather_exchange": 1000.0 * t_s["gather_exchange"] / psteps,
                     "ms_gather_d2h": 1000.0 * t_s["gather_d2h"] / psteps, "ms_host_waiting_for_kernels": 1000.0 * t_s["sync_wait"] / psteps,
                     "ms_rank_rank0": float(np.mean([x["ms_rank"] for x in st_s])), "ms_align_rank0": float(np.mean([x["ms_align"] for x in st_s])),
                     "ms_rank_setup_rank0": float(np.mean([x["ms_rank_setup"] for x in st_s])),
                     "gathered_hits_per_step": int(len(got[0])) if got else 0, "peer_errors": [str(e) for e in perr],
                     "note": "one GPU: rank 0's kernels run alone (peers only gather), the %d-rank exchange is device-to-device copies instead of "
                             "xGMI transfers (7 x ~%d KB in parallel over separate links at ~153 GB/s each: < 0.2 ms); the gather of step i runs beside "
                             "the kernels of step i+1 (enqueued before the host waits for step i)" % (N, int(len(got[0]) * 80 / N / 1000) if got else 0)}
            for pb in peers + small:
                pb.close()
            for a in gbuf:
                capi.lib().ugs_host_unregister(a.ctypes.data)
            gbuf = None
            for c in comms:
                c.close()
        # ---- the other named configurations, once each (driver-visible)
        which = [] if args.other_configs == "none" else (["C5", "C4", "C3", "ID90"] if args.other_configs == "auto" else [x for x in args.other_configs.split(",") if x])
        if which:
            for b in bats:
                b.close()
            gdb.close()
            others = []
            for name in which:
                try:
                    others.append(other_config(capi, synth, name, local_rank, parity=args.parity_sample > 0))
                except Exception as e:                              # (reported, never hidden: the C2 line above is already measured)
                    others.append({"workload": name, "error": "%s: %s" % (type(e).__name__, e)})

    if rank == 0:
        st = stats[-1]
        ms_rank = float(np.mean([s["ms_rank"] for s in stats]))
        ms_align = float(np.mean([s["ms_align"] for s in stats]))
        ms_setup = float(np.mean([s["ms_rank_setup"] for s in stats]))
        # algorithmic bytes per launch: B(q) = 4*P(q) + L_q + sum L_candidates
        b_rank = 4 * st["postings"] + st["query_letters"]
        b_align = st["query_letters"] + st["target_letters"]
        # the ranking stage is the bitmap kernel k_rank2 (ugs_rank2.hip) followed by k_rank over the units k_rank2 deferred (HIP events
        # around each on the handle's stream); the roofline of k_rank2 counts the algorithmic bytes of the units it ranked itself
        r2_on = bool(khits) and all(k["r2_launched"] for k in khits)
        ms_rank2 = float(np.mean([k["ms_rank2"] for k in khits])) if r2_on else 0.0
        ms_rank_def = float(np.mean([k["ms_rank_deferred"] for k in khits])) if r2_on else 0.0
        units_step = max(1, khits[-1]["r2_units"] + khits[-1]["deferred"]) if khits else 1
        r2_share = (khits[-1]["r2_units"] / units_step) if r2_on else 0.0
        if r2_on and ms_rank2 >= ms_align:
            dom, b_dom, ms_dom = "k_rank2", b_rank * r2_share, ms_rank2
        elif ms_rank >= ms_align:
            dom, b_dom, ms_dom = "k_rank", b_rank, ms_rank
        else:
            dom, b_dom, ms_dom = "k_align", b_align, ms_align
        # HBM traffic per launch from the PMC passes of the same command (profiles/*_pmc.json, FETCH_SIZE
        # KiB x2 gfx950 correction + WRITE_SIZE); only attached when the workload shape matches
        traffic = None
        mix = {}
        traffic_source = None
        pmcs = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc.json")), reverse=True)   # newest round first
        for pmc_name in pmcs:
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", pmc_name)))
                if pm.get("db_seqs") == db.n and pm.get("queries") == qs.n and dom in pm.get("traffic_bytes_per_launch", {}):
                    if pm.get("csrc_sha16") != csrc_sha:        # counters of ANOTHER build say nothing about this one: no traffic rather than a stale ratio
                        traffic_source = traffic_source or ("none: the newest PMC passes of this shape (profiles/%s) were taken on source hash %s, this build is %s" %
                                                            (pmc_name, pm.get("csrc_sha16"), csrc_sha))
                        continue
                    traffic = pm["traffic_bytes_per_launch"][dom]
                    traffic_source = "profiles/" + pmc_name + " (rocprofv3 --pmc passes of this command on this same source hash, not measured in this run)"
                    mix = pm.get("instruction_mix_per_launch", {})
                    break
            except (OSError, ValueError):
                pass

        def issue_roofline(kernel, ms):
            """VALU-issue ceiling of an integer/LDS kernel: wave-level VALU instructions of one launch (SQ_INSTS_VALU, PMC pass of
            this command in profiles/) over the live kernel time, against 256 CU x 4 SIMD x 2.4 GHz / 2 cycles per wave64 VALU op
            (MI355X_MICROARCH.md).  The instruction count is a property of the code and the workload, not of the run."""
            m = mix.get(kernel)
            if not m or "SQ_INSTS_VALU" not in m:
                return None
            peak = 256 * 4 * 2.4e9 / 2
            ach = m["SQ_INSTS_VALU"] / (ms * 1e-3)
            out = {"bound": "valu-issue", "valu_insts_per_launch": m["SQ_INSTS_VALU"], "achieved_Ginst_s": ach / 1e9, "peak_Ginst_s": peak / 1e9,
                   "frac": ach / peak}
            if "SQ_INSTS_LDS" in m:
                out["lds_insts_per_launch"] = m["SQ_INSTS_LDS"]
            if "SQ_WAVE_CYCLES" in m and "SQ_WAIT_INST_ANY" in m:
                out["wave_time_waiting_frac"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
            return out
        achieved = b_dom / (ms_dom * 1e-3) / 1e9
        n_hits = n_hits_main
        # (the CPU baseline is a single-GPU-run item - rank 0 at N=1, measured above: the other ranks of a multi-GPU run would only wait for it)
        how = ("ugs_gather_results (libugs_rccl.so: ncclAllGather of sizes + grouped ncclSend/ncclRecv), issued beside the next step's kernels"
               if comm is not None else ("host transport: every rank fetches, the tables travel through the rank group's sockets (dry run, not a measurement)" if group is not None else "none (one GPU: plain fetch)"))
        if workload == "C2":
            wl = ("C2: usearch_global %d x %d nt queries per GPU and step vs %d-seq DB, -id %.2f -strand plus, reference defaults (maxaccepts 1, "
                  "maxrejects 32, Big ranking path); every step uploads and searches a batch different from the previous one%s" %
                  (shard_n, args.length, db.n, args.id, "" if world == 1 else "; %d GPUs = %d queries per step, DB replicated, one gather of the hit tables to rank 0 per step" % (world, total_q)))
            if strong_c2:
                wl = "STRONG scaling, " + wl.replace("per GPU and step", "per GPU (= %d per step over all GPUs, contiguous shards)" % total_q)
        else:
            wl = ("C4: usearch_global %d x %d nt queries in %d contiguous shard(s) vs a %d-seq DB replicated per GPU, -id %.2f "
                  "-strand plus, reference defaults; one RCCL gather of the hit tables to rank 0 per step" %
                  (total_q, args.length, world, db.n, args.id))
        line = {
            "metric": "query-seqs/s usearch_global -id 0.97 (search phase: query batch on the host -> hit table on the host)",
            "value": value, "unit": "query-seqs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / steps, "higher_is_better": True, "scaling": "weak" if (workload == "C2" and not strong_c2) else "strong",
            "vs_baseline": None, "dtype": "u8/u32 (int32 half-unit DP scores)", "data": "synthetic",
            "config": {"workload": wl, "queries_per_step": total_q, "queries_per_gpu": shard_n, "db_seqs": db.n, "seq_len": args.length,
                       "parallelism": "query shards, DB replicated per GPU" if world > 1 else "single GPU", "gather": how},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": b_dom, "kernel_ms": ms_dom,
                         "bytes_per_query": b_dom / max(qs.n, 1)},
            "roofline_per_kernel": {
                "k_rank2": ({"bound": "hbm", "algorithmic_bytes_per_launch": b_rank * r2_share, "kernel_ms": ms_rank2,
                             "achieved_GBps": b_rank * r2_share / (ms_rank2 * 1e-3) / 1e9, "frac": b_rank * r2_share / (ms_rank2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "units_ranked": khits[-1]["r2_units"], "units_deferred_to_k_rank": khits[-1]["deferred"],
                             "k_rank_over_deferred_units_ms": ms_rank_def,
                             "issue_roofline": issue_roofline("k_rank2", ms_rank2)} if r2_on else None),
                "k_rank": {"note": "ranking stage as a whole: k_rank2 + k_rank over its deferred units" if r2_on else "the ranking kernel",
                           "bound": "hbm", "algorithmic_bytes_per_launch": b_rank, "kernel_ms": ms_rank,
                           "achieved_GBps": b_rank / (ms_rank * 1e-3) / 1e9, "frac": b_rank / (ms_rank * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "issue_roofline": issue_roofline("k_rank", ms_rank)},
                "k_rank_setup": {"bound": "latency (dependent loads, one wavefront per query)", "kernel_ms": ms_setup},
                "k_align": {"bound": "valu-issue (integer ALU / LDS, not a bandwidth kernel)", "algorithmic_bytes_per_launch": b_align,
                            # SURVEY.md 8(d)'s rule, one byte per letter of the query and of every candidate visited (the synthetic targets all
                            # have the query length); algorithmic_bytes_per_launch above = the bytes k_align fetches (packed planes)
                            "algorithmic_bytes_per_launch_8d": st["query_letters"] + st["pairs_aligned"] * (st["query_letters"] // max(qs.n, 1)),
                            "kernel_ms": ms_align, "achieved_GBps": b_align / (ms_align * 1e-3) / 1e9,
                            "pair_alignments_per_s": st["pairs_aligned"] / (ms_align * 1e-3),
                            "valu_per_pair": (mix["k_align"]["SQ_INSTS_VALU"] / max(st["pairs_aligned"], 1)) if "k_align" in mix else None,
                            "issue_roofline": issue_roofline("k_align", ms_align)}},
            "cpu_baseline": cb,
            # one entry per configuration that was compared with the reference BINARY's own output on this box (C2: 384 k queries, C5: 100 k
            # protein queries, C3: the -uc of a 150 k-read prefix at -threads 1)
            "parity_sample": ([dict(parity, config="C2")] if parity else []) +
                             [dict(o["parity_sample"], config=o["workload"].split(":")[0]) for o in (others or []) if isinstance(o, dict) and o.get("parity_sample")],
            "detail": {"csrc_sha16": csrc_sha, "runtime_libs": all_libs, "other_configs": others, "strong_scaling_proxy": proxy,
                       "ms_rank": ms_rank, "ms_rank_setup": ms_setup, "ms_align": ms_align, "hits_per_step": n_hits,
                       "postings_per_query": st["postings"] / max(qs.n, 1),
                       "pairs_aligned_per_query": st["pairs_aligned"] / max(qs.n, 1),
                       "dp_gcells_per_s": st["dp_cells"] / max(ms_align * 1e-3, 1e-9) / 1e9,
                       "host_ms_search_sync": 1000.0 * t_main["search_sync"] / steps,
                       "host_ms_upload_issue": 1000.0 * t_main["upload_issue"] / steps,
                       "host_ms_fetch": 1000.0 * t_main["fetch"] / steps,
                       "host_ms_gather": 1000.0 * t_main["gather"] / (steps + 1),
                       "host_ms_sync_wait_after_collect": 1000.0 * t_main["sync_wait"] / steps,
                       "per_rank": per_rank,
                       "index_build_s": t_index, "first_upload_search_s": t_upload, "gen_s": t_gen,
                       "db_hbm_bytes": db_hbm_bytes,
                       "gpu_over_cpu": (value / world / cb["value"]) if cb else None},
        }
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if comm is not None:
        comm.close()
    if group is not None:
        group.barrier()
        group.close()


if __name__ == "__main__":
    main()
