#!/bin/bash
# round 3 GPU call 3: quad-shaped gathers + range guard -- parity, micro-bench, bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_dense.py tests/test_gpu_pipeline.py tests/test_gpu_train_ops.py -x -q -m gpu > gpurun_out/t3.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/t3.log
FRAMES=48 timeout 600 python tools/conv_bench.py sparse f16x2 10 > gpurun_out/cb_quad.txt 2>&1
cat gpurun_out/cb_quad.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras > gpurun_out/bench_r03b.json 2> gpurun_out/bench_r03b.err
echo "bench rc $?"; tail -c 400 gpurun_out/bench_r03b.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r03b.json'))
print(d["value"], d["ms_per_step"])
for k,v in d["roofline"]["all_conv_kernels"].items(): print("%-40s %.1f TF  %.4f ms/frame"%(k,v['tflops'],v['ms_per_frame']))
PY
