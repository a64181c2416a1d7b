#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_dense.py tests/test_gpu_pipeline.py tests/test_gpu_train_ops.py tests/test_gpu_train.py tests/test_gpu_autograd.py -x -q -m gpu > gpurun_out/t5.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/t5.log
FRAMES=48 timeout 600 python tools/conv_bench.py sparse f16x2 10 2>&1 | grep -v amdgpu.ids > gpurun_out/cb_buf.txt
cat gpurun_out/cb_buf.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras > gpurun_out/bench_r03c.json 2> gpurun_out/bench_r03c.err
echo "bench rc $?"; tail -c 400 gpurun_out/bench_r03c.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r03c.json'))
print(d["value"], d["ms_per_step"], d["config"]["range_guard"][-20:])
for k,v in d["roofline"]["all_conv_kernels"].items(): print("%-40s %.1f TF  %.4f ms/frame"%(k,v['tflops'],v['ms_per_frame']))
print({k: round(v["us_per_frame"],1) for k,v in d["hbm_stages"].items() if isinstance(v, dict)})
PY
