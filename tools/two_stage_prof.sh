#!/bin/bash
# rocprofv3 --kernel-trace --stats of tools/two_stage_bench.py (FRAMES frames per step): per-kernel table -> gpurun_out/two_stage_kernel_stats.csv
R=$PWD; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/two_stage_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/two_stage_prof -- python $R/tools/two_stage_bench.py > $R/gpurun_out/two_stage_under_rocprof.txt 2>/dev/null
cd $R
cp $(ls gpurun_out/two_stage_prof/*/*kernel_stats.csv | head -1) gpurun_out/two_stage_kernel_stats.csv
cat gpurun_out/two_stage_under_rocprof.txt
python - <<'PY'
import csv, re
rows = list(csv.DictReader(open("gpurun_out/two_stage_kernel_stats.csv")))
steps = 10.0      # 2 warm + 4 timed two-stage + 4 first-stage-only forwards
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms over the run: %.1f" % (tot / 1e6))
for r in rows[:70]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*$", "", n)
    print("%-70s calls %6s avg us %8.1f total ms %8.2f" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
