cd $GRAFT_REPO_ROOT
RQ_SHAPE=aa python tools/rank_quick.py 1000000 2>&1 | grep -v "^\[ugs\]"
python tools/rank_quick.py 1000000 2>&1 | grep -v "^\[ugs\]"
python tools/rank_quick.py 1000000 2>&1 | grep -v "^\[ugs\]"
python -m pytest tests/test_gpu_paths.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | grep -E "passed|failed|^E " | tail -5
