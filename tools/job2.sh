#!/bin/bash
# round 3 GPU call 2: gather-shape probe, new tests, remaining PMC passes, bench with extras
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 tools/probe/gather_probe > gpurun_out/gather_probe.txt 2>&1
echo "probe rc $?"
timeout 900 python -m pytest tests/test_anchor_head.py tests/test_prefilter.py tests/test_gpu_train_ops.py tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/t2.log 2>&1
echo "pytest rc $?"; tail -5 gpurun_out/t2.log
PASSES="ta1 ta2 td tcc fetch write derived" PASS_TIMEOUT=200 bash tools/pmc_rowwave.sh r03 > gpurun_out/pmc_rw_r03b.log 2>&1
echo "pmc rc $?"
timeout 600 python bench.py > gpurun_out/bench_r03a.json 2> gpurun_out/bench_r03a.err
echo "bench rc $?"; tail -c 600 gpurun_out/bench_r03a.err
