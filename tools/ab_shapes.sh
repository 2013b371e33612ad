# the other shapes of k_rank under a variant build: mid-identity usearch_local (8-bit BATCH kernels), C3 cluster_fast (LONG + small path), C1/C4/C5
for v in "$@"; do
  export UGS_LIB=usearch12_amd/variants/libugs_$v.so
  [ "$v" = "intree" ] && unset UGS_LIB
  echo "== $v"
  python tools/local_bench.py --db 1000000 --id 0.9 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' local nt ms_rank', round(d['ms_rank'],1), 'hits', d['hits'])"
  python tools/cluster_bench.py --reads 5000000 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' C3 s', round(d['seconds'],2), 'ms_rank', round(d['ms_rank']), 'clusters', d['n_clusters'])"
  python tools/gpu_config_check.py 2>/dev/null | grep config | python -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print(' ', d['config'][:12], 'ms_rank', d['ms_rank'], 'ok', d['ok'])"
done
