cd $GRAFT_REPO_ROOT
python tools/rank_quick.py 1000000 2>&1 | grep -v "^\[ugs\]"
UGS_LIB=usearch12_amd/variants/libugs_aold.so python tools/rank_quick.py 1000000 2>&1 | grep -v "^\[ugs\]"
RQ_SHAPE=aa python tools/rank_quick.py 1000000 2>&1 | grep -v "^\[ugs\]"
RQ_SHAPE=aa UGS_LIB=usearch12_amd/variants/libugs_aold.so python tools/rank_quick.py 1000000 2>&1 | grep -v "^\[ugs\]"
