#!/usr/bin/env python3
"""Golden for cluster_fast ACROSS THE NATURAL small -> Big LATCH (udbusortedsearcher.cpp:39-58 at the default -big 100000;
VERDICT r02 item 7): 700 000 reads of the C3 model (usearch12_amd.synth.make_reads(3, 700_000, n_species=50_000): 50 k species,
Pareto(1.2) abundances, 1 % substitutions, 0.1 % / 0.1 % indels, -id 0.97) through the compiled, UNMODIFIED reference
(oracle/_ref/usearch12 -cluster_fast ... -threads 1; about four minutes).  The outputs are too large to commit (40 MB); what
is committed is cluster_big_manifest.json: sha256 of the reference's -uc and -centroids files, the cluster count, the sha256
of the generated reads and the command line.  The GPU test regenerates the reads, clusters them on the device, writes both
files through the product's writers and compares the digests."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from usearch12_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "usearch12")
CASE = dict(seed=3, n=700_000, species=50_000, id=0.97, strand="plus")


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    assert os.path.exists(REF), "build the reference first: oracle/build_ref.sh"
    c = dict(CASE)
    r = synth.make_reads(c["seed"], c["n"], n_species=c["species"])
    h = hashlib.sha256()
    h.update(r.offs.tobytes())
    h.update(r.seqs.tobytes())
    with tempfile.TemporaryDirectory() as tmp:
        fa, uc, cen = (os.path.join(tmp, x) for x in ("reads.fa", "o.uc", "o.cent.fa"))
        r.write_fasta(fa)
        cmd = [REF, "-cluster_fast", fa, "-id", str(c["id"]), "-uc", uc, "-centroids", cen, "-threads", "1", "-strand", c["strand"]]
        t0 = time.time()
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        c["reference_seconds"] = round(time.time() - t0, 1)
        c["uc_sha256"], c["centroids_sha256"] = sha(uc), sha(cen)
        c["uc_bytes"], c["centroids_bytes"] = os.path.getsize(uc), os.path.getsize(cen)
        c["n_clusters"] = sum(1 for ln in open(cen) if ln.startswith(">"))
        c["n_uc_records"] = sum(1 for _ in open(uc))
    c["reads_sha256"] = h.hexdigest()
    c["cmd"] = "usearch12 -cluster_fast reads.fa -id 0.97 -uc o.uc -centroids o.cent.fa -threads 1 -strand plus"
    with open(os.path.join(HERE, "cluster_big_manifest.json"), "w") as f:
        json.dump({"cl_c3_natural_latch": c}, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps(c))


if __name__ == "__main__":
    main()
