cd $GRAFT_REPO_ROOT
for e in 0 1; do
  if [ $e = 1 ]; then export UGS_BATCH_STREAMS=1; fi
  python bench.py --steps 20 --warmup 5 --cpu-baseline none --other-configs none --emulate-world 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams=$e', round(d['value']/1e6,2), round(d['ms_per_step'],2), round(d['detail']['ms_rank'],2), round(d['detail']['ms_align'],2), d['detail']['hits_per_step'])"
done
