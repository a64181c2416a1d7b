#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train.py tests/test_gpu_train_dist.py tests/test_gpu_sparse.py tests/test_gpu_dense.py tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/t7.log 2>&1
echo "pytest rc $?"; tail -5 gpurun_out/t7.log
for f in 0 1; do
  CPD_TRAIN_FUSE_BN=$f timeout 300 python bench.py --mode train --steps 40 --warmup 10 --no-roofline > gpurun_out/train_fuse$f.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/train_fuse$f.json')); print('fuse $f', round(d['ms_per_step'],3), 'ms/step')"
done
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-roofline > gpurun_out/bench_r03d.json 2> gpurun_out/bench_r03d.err
python -c "import json; d=json.load(open('gpurun_out/bench_r03d.json')); print('infer', round(d['value'],1), round(d['ms_per_step'],2))"
