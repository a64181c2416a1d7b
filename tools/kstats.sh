#!/bin/bash
# usage: tools/kstats.sh <tag> <steps_total> <bench args...>: rocprofv3 --kernel-trace --stats of `python bench.py <args>`;
# prints the per-kernel table per step (steps_total = warmup + steps of that run) and keeps csv + bench line in gpurun_out/<tag>*
R=$PWD; tag=$1; steps=$2; shift 2
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -- python $R/bench.py "$@" > $R/gpurun_out/${tag}_under_rocprof.json 2>/dev/null
cd $R
cp $(ls gpurun_out/${tag}_prof/*/*kernel_stats.csv | head -1) gpurun_out/${tag}_kernel_stats.csv
python tools/prof_summary.py gpurun_out/${tag}_kernel_stats.csv gpurun_out/${tag}_under_rocprof.json $steps 45
