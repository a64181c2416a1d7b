R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ts_prof
FRAMES=16 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ts_prof -- python $R/tools/two_stage_bench.py > $R/gpurun_out/ts_bench.txt 2>&1
cd $R
cp $(ls gpurun_out/ts_prof/*/*kernel_stats.csv | head -1) gpurun_out/ts_kernel_stats.csv
cp $(ls gpurun_out/ts_prof/*/*kernel_trace.csv | head -1) gpurun_out/ts_kernel_trace.csv
