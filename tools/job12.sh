#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for f in 2 4 8; do
  timeout 600 python bench.py --mode train --frames $f --steps 20 --warmup 6 --no-roofline > gpurun_out/r03_train_bench_${f}frames.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/r03_train_bench_${f}frames.json')); print('train frames $f:', round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms/step')"
done
