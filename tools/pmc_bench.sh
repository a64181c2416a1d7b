#!/bin/bash
# Round profile set for the default bench (run on the GPU box through gpurun, from the repo root):
#   gpurun_out/r01_pmc_summary.json         three separate --pmc passes (FETCH_SIZE | WRITE_SIZE | MFMA busy), done FIRST and
#                                           copied to profiles/ so that the bench lines below carry the fresh `traffic`
#   gpurun_out/r01_bench.json               python bench.py
#   gpurun_out/r01_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/r01_bench_under_rocprof.json bench line of that profiled run
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/pmc
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc/fetch -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc/write -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $R/gpurun_out/pmc/mfma -o p -- $CMD > /dev/null 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/r01_pmc_summary.json
cp gpurun_out/r01_pmc_summary.json profiles/r01_pmc_summary.json
python bench.py > $R/gpurun_out/r01_bench.json 2> $R/gpurun_out/r01_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/r01_bench_under_rocprof.json 2>/dev/null
cp $(ls $R/gpurun_out/prof_stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/r01_bench_kernel_stats.csv
cd $R
