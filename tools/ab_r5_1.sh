cd $GRAFT_REPO_ROOT
python tools/rank_quick.py 1000000 2>&1 | grep -v "^\[ugs\]"
UGS_R2_CLCAP=208 UGS_R2_KCAP=252 python tools/rank_quick.py 1000000 2>&1 | grep -v "^\[ugs\]"
python -m pytest tests/test_gpu_paths.py tests/test_gpu_parity.py tests/test_gpu_cluster.py -x -q -m gpu 2>&1 | tail -3
