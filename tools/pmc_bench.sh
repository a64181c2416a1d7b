#!/bin/bash
# Round profile set for the default bench (run on the GPU box through gpurun, from the repo root):  tools/pmc_bench.sh r02
#   profiles/<tag>_pmc_summary.json          three separate --pmc passes (FETCH_SIZE | WRITE_SIZE | MFMA busy), done FIRST so
#                                            that the bench lines below carry the fresh `traffic`
#   profiles/<tag>_bench.json                python bench.py
#   profiles/<tag>_bench_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same command
#   profiles/<tag>_bench_under_rocprof.json  bench line of that profiled run
tag=${1:-r02}
R=$PWD
mkdir -p $R/gpurun_out $R/profiles
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/pmc
# one stream, no extras: per-kernel averages of the headline configuration only (the bench line's own roofline pass is single-stream too)
CMD="python $R/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline --no-extras"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc/fetch -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc/write -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $R/gpurun_out/pmc/mfma -o p -- $CMD > /dev/null 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/${tag}_pmc_summary.json > /dev/null
cp gpurun_out/${tag}_pmc_summary.json profiles/${tag}_pmc_summary.json
python bench.py > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python $R/bench.py --streams 1 --no-cpu-baseline --no-extras > $R/gpurun_out/${tag}_bench_under_rocprof.json 2>/dev/null
cp $(ls $R/gpurun_out/prof_stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/${tag}_bench_kernel_stats.csv
cd $R
python tools/prof_summary.py gpurun_out/${tag}_bench_kernel_stats.csv gpurun_out/${tag}_bench_under_rocprof.json auto 48
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench.json")); r = d["roofline"]
print("bench: %.1f frames/s, %.2f ms/step; dominant %s: %.1f TF = %.3f of %.0f, launch %.1f us, traffic %s; path_hbm_frac %.3f" %
      (d["value"], d["ms_per_step"], r["kernel"], r["achieved"], r["frac"], r["peak"], r["avg_launch_us"], r["traffic"], r.get("path_hbm_frac", -1)))
print("cpu_baseline", d.get("cpu_baseline"))
PY
# variants quoted in DESIGN.md §5 (one bench line each; no CPU baseline, no roofline pass)
if [ "${2:-}" = "variants" ]; then
  for v in "bf16x3:--conv-math bf16x3" "f32:--conv-math f32" "streams1:--streams 1" "streams3:--streams 3" "device_results:--device-results" "1frame:--frames 1 --steps 100 --streams 1" "4frames:--frames 4 --steps 100 --streams 1" "4frames_2streams:--frames 4 --steps 100" "16frames:--frames 16 --steps 48" "canonical_rows:--row-order canonical" "fp32_dense_maps:--dense-pairs 0" "modules:--api modules"; do
    python bench.py --no-cpu-baseline --no-roofline --no-extras ${v#*:} > gpurun_out/${tag}_bench_${v%%:*}.json 2>> gpurun_out/${tag}_bench.err
    python -c "import json,sys; d=json.load(open('gpurun_out/${tag}_bench_${v%%:*}.json')); print('${v%%:*}', round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms/step')"
  done
  python bench.py --mode train --steps 40 --warmup 10 > gpurun_out/${tag}_train_bench.json 2>> gpurun_out/${tag}_bench.err
  python -c "import json; d=json.load(open('gpurun_out/${tag}_train_bench.json')); print('train', round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms/step')"
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train -- python $R/bench.py --mode train --steps 40 --warmup 10 > $R/gpurun_out/${tag}_train_bench_under_rocprof.json 2>/dev/null
  cp $(ls $R/gpurun_out/prof_train/*/*kernel_stats.csv | head -1) $R/gpurun_out/${tag}_train_kernel_stats.csv
  cd $R
fi
