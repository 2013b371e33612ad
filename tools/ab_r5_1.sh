cd $GRAFT_REPO_ROOT
python tools/rank_quick.py 1000000 2>&1 | grep -v "^\[ugs\]" | cut -c1-100
UGS_LIB=usearch12_amd/variants/libugs_base.so python tools/rank_quick.py 1000000 2>&1 | grep -v "^\[ugs\]" | cut -c1-100
python tools/rank_quick.py 1000000 2>&1 | grep -v "^\[ugs\]" | cut -c1-100
UGS_LIB=usearch12_amd/variants/libugs_base.so python tools/rank_quick.py 1000000 2>&1 | grep -v "^\[ugs\]" | cut -c1-100
