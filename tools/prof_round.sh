#!/bin/bash
# final-state profiles of a round (run on the GPU box from the repository root): bench line, rocprofv3 kernel stats of the same
# command, PMC passes (FETCH_SIZE and WRITE_SIZE each in a pass of its own: together they exceed what the hardware collects and
# rocprofv3 aborts; every pass under `timeout`)
tag=${1:-r02d}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
if [ "$2" != "pmc-only" ]; then
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_c2.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats -- python $GRAFT_REPO_ROOT/bench.py --cpu-baseline none --other-configs none --emulate-world 0 2>/dev/null | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_c2_under_rocprof.json )
f=$(find gpurun_out/${tag}_stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_kernel_stats.csv; python tools/prof_summary.py gpurun_out/${tag}_kernel_stats.csv 14 > gpurun_out/${tag}_kernel_stats.txt
rm -rf gpurun_out/${tag}_stats
fi
timeout 200 tools/pmc_pass.sh p1 FETCH_SIZE > gpurun_out/${tag}_pmc_pass1.txt 2>&1
timeout 200 tools/pmc_pass.sh p2 WRITE_SIZE > gpurun_out/${tag}_pmc_pass2.txt 2>&1
timeout 200 tools/pmc_pass.sh p3 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES > gpurun_out/${tag}_pmc_pass3.txt 2>&1
timeout 200 tools/pmc_pass.sh p4 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES > gpurun_out/${tag}_pmc_pass4.txt 2>&1
rm -rf gpurun_out/pmc_p1 gpurun_out/pmc_p2 gpurun_out/pmc_p3 gpurun_out/pmc_p4
cat gpurun_out/${tag}_pmc_pass*.txt | tail -40
