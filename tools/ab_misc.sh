# k_xdrop / k_local / k_inbatch under variant builds (tools/build_variant_of.sh): kernel ms per variant
for v in intree xdrop_m xdrop_r xdrop_mr; do
  export UGS_LIB=usearch12_amd/variants/libugs_$v.so; [ "$v" = "intree" ] && unset UGS_LIB
  python tools/xdrop_bench.py --cpu-sample 200 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', {k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if 'per_s' in k or k.startswith('ms')})"
done
for v in intree local_m local_r local_mr; do
  export UGS_LIB=usearch12_amd/variants/libugs_$v.so; [ "$v" = "intree" ] && unset UGS_LIB
  python tools/local_bench.py --db 1000000 --id 0.9 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v nt ms_local', round(d['ms_local'],2), d['hits'])"
  python tools/local_bench.py --db 1000000 --aa --len 300 --id 0.8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v aa ms_local', round(d['ms_local'],2), 'ms_rank', round(d['ms_rank'],2), d['hits'])"
done
for v in intree inbatch_m inbatch_r; do
  export UGS_LIB=usearch12_amd/variants/libugs_$v.so; [ "$v" = "intree" ] && unset UGS_LIB
  python tools/cluster_bench.py --reads 5000000 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v C3 s', round(d['seconds'],2), 'inbatch', d['host_s']['s_inbatch'], 'clusters', d['n_clusters'])"
done
