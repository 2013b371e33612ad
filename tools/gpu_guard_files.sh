#!/bin/bash
# tools/gpu_guard_files.sh <tag> [test files...]  - every GPU test FILE in a pytest process of its own under the guard allocator
# (UGS_GUARD_ALLOC=1: a buffer overrun of any kernel is a GPU memory fault, which kills the process), so that one faulting file does not
# hide the others.  Per file: gpurun_out/<tag>_<file>.{out,err}; the err log holds the runtime's "Memory access fault ... on address X",
# the aborting thread's backtrace and the guard allocator's buffer table (UGS_ABORT_BT=stderr) - the buffer that ends at page X is the one
# that was overrun, its allocation site is in the table.  UGS_DEBUG_SYNC=1 logs the stage of a search that was running.
tag=$1; shift
files=("$@")
if [ ${#files[@]} -eq 0 ]; then files=(tests/test_gpu_*.py tests/test_udb.py tests/test_zz_gpu_coverage.py); fi
mkdir -p gpurun_out
export UGS_GUARD_ALLOC=${UGS_GUARD_ALLOC:-1} UGS_ABORT_BT=stderr HSA_ENABLE_IPC_MODE_LEGACY=0 UGS_DEBUG_SYNC=${UGS_DEBUG_SYNC:-1}
: > gpurun_out/${tag}_summary.txt
for f in "${files[@]}"; do
  b=$(basename $f .py)
  t0=$(date +%s)
  timeout ${FILE_TIMEOUT:-900} python -m pytest $f -m gpu -q -p no:cacheprovider --capture=sys ${SUITE_ARGS} > gpurun_out/${tag}_$b.out 2> gpurun_out/${tag}_$b.err
  rc=$?
  t1=$(date +%s)
  echo "$b rc=$rc seconds=$((t1-t0)) : $(tail -1 gpurun_out/${tag}_$b.out | cut -c1-150)" >> gpurun_out/${tag}_summary.txt
  if [ $rc -ne 0 ]; then grep -h "Memory access fault" gpurun_out/${tag}_$b.err | head -3 >> gpurun_out/${tag}_summary.txt; fi
done
cat gpurun_out/${tag}_summary.txt
