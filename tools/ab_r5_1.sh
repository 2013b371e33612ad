cd $GRAFT_REPO_ROOT
python tools/rank_quick.py 1000000 2>&1 | grep -v "^\[ugs\]"
UGS_LIB=usearch12_amd/variants/libugs_tb0.so python tools/rank_quick.py 1000000 2>&1 | grep -v "^\[ugs\]"
RQ_SHAPE=aa python tools/rank_quick.py 1000000 2>&1 | grep -v "^\[ugs\]"
RQ_SHAPE=aa UGS_LIB=usearch12_amd/variants/libugs_tb0.so python tools/rank_quick.py 1000000 2>&1 | grep -v "^\[ugs\]"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_edges.py -q -m gpu -x 2>&1 | grep -E "passed|failed|^E " | tail -5
