import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from usearch12_amd import capi, synth
r = synth.make_reads(3, 5000000, n_species=50000, length=300)
p = capi.cluster_params(0.97)
res = capi.UgsCluster(p, r.seqs, r.offs)
seeds = res.uniq_seed[res.centroid_uniq]
lens = np.diff(r.offs.astype(np.int64))
offs = np.zeros(len(seeds) + 1, np.uint64); offs[1:] = np.cumsum(lens[seeds])
seqs = np.concatenate([r.seqs[int(r.offs[i]):int(r.offs[i + 1])] for i in seeds[:]])
print("centroids", len(seeds), flush=True)
gdb = capi.UgsDB(p, seqs, offs, device=0)
q = r.slice(4900000, 4900000 + 16384)
bat = capi.UgsBatch(gdb, q.n, int(q.offs[-1]))
bat.upload(q.seqs, q.offs)
os.environ["UGS_PHASE_CLOCKS"] = "1"
for _ in range(2):
    bat.search(); bat.sync(); st = bat.stats()
print({k: st[k] for k in ("ms_rank", "ms_align", "ms_rank_setup", "postings", "pairs_aligned")})
