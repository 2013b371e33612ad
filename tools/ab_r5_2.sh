cd $GRAFT_REPO_ROOT
for i in 1 2; do
python tools/rank_quick.py 1000000 2>&1 | grep -v "^\[ugs\]" | cut -c1-100
UGS_LIB=usearch12_amd/variants/libugs_base.so python tools/rank_quick.py 1000000 2>&1 | grep -v "^\[ugs\]" | cut -c1-100
done
RQ_SHAPE=aa python tools/rank_quick.py 300000 2>&1 | grep -v "^\[ugs\]" | cut -c1-100
RQ_SHAPE=aa UGS_LIB=usearch12_amd/variants/libugs_base.so python tools/rank_quick.py 300000 2>&1 | grep -v "^\[ugs\]" | cut -c1-100
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -5
