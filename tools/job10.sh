#!/bin/bash
# round 3 final profile set
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/pmc_bench.sh r03 variants > gpurun_out/pmc_bench_r03.log 2>&1
tail -40 gpurun_out/pmc_bench_r03.log
PASS_TIMEOUT=200 bash tools/pmc_rowwave.sh r03after > gpurun_out/pmc_rw_r03after.log 2>&1
tail -3 gpurun_out/pmc_rw_r03after.log | cut -c1-600
