#!/bin/bash
# Inference bench (default config) per-layer table + rocprofv3 kernel stats.
mkdir -p gpurun_out
python bench.py --no-cpu-baseline --layers > gpurun_out/infer_bench.json 2> gpurun_out/infer_layers.txt
cat gpurun_out/infer_layers.txt | grep -v Warn
python -c "
import json; d=json.load(open('gpurun_out/infer_bench.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['avg_launch_us'], r['chip_conv_tflops'])
for k,v in r['all_conv_kernels'].items(): print(k, v)
"
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/infer_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/infer_prof -- python $R/bench.py --no-cpu-baseline --no-roofline > /dev/null 2>&1
cd $R
f=$(ls gpurun_out/infer_prof/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/infer_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
nf = 20 * 8
print("total kernel ms / frame: %.3f" % (tot / nf / 1e6))
for r in rows[:30]:
    print("%-60s calls %5s  avg %9.1f us  %5.1f%%  us/frame %.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3,
          float(r["Percentage"]), float(r["TotalDurationNs"]) / nf / 1e3))
PY
