#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -x -q -m gpu > gpurun_out/t18.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/t18.log
bash tools/pmc_bench.sh r03 variants > gpurun_out/pmc_bench_r03.log 2>&1
tail -16 gpurun_out/pmc_bench_r03.log
PASS_TIMEOUT=200 bash tools/pmc_rowwave.sh r03after > gpurun_out/pmc_rw_r03after.log 2>&1
for f in 2 4 8; do
  timeout 600 python bench.py --mode train --frames $f --steps 20 --warmup 6 --no-roofline > gpurun_out/r03_train_bench_${f}frames.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/r03_train_bench_${f}frames.json')); print('train frames $f:', round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms/step')"
done
