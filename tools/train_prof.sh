#!/bin/bash
# Train-step bench + rocprofv3 kernel stats (run on the GPU box through gpurun).
mkdir -p gpurun_out
python bench.py --mode train --steps 8 --warmup 4 > gpurun_out/train_bench.json 2> gpurun_out/train_bench.err
cat gpurun_out/train_bench.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/train_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/train_prof -- python $R/bench.py --mode train --steps 8 --warmup 4 --no-roofline > $R/gpurun_out/train_bench_under_rocprof.json 2>/dev/null
cd $R
f=$(ls gpurun_out/train_prof/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/train_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms / step: %.2f" % (tot / 12e6))
for r in rows[:16]:
    print("%-70s calls %5s  avg %9.1f us  %5.1f%%  ms/step %.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3,
          float(r["Percentage"]), float(r["TotalDurationNs"]) / 12e6))
PY
