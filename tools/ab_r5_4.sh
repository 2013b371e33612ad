cd $GRAFT_REPO_ROOT
for v in r2c1 r2c2; do echo == aa $v; UGS_PHASE_CLOCKS=1 RQ_SHAPE=aa UGS_LIB=usearch12_amd/variants/libugs_$v.so python tools/rank_quick.py 300000 2>&1 | cut -c1-400; done
