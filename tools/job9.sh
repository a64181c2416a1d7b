#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CPD_GC_ROWWAVE_DEEP=1 timeout 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/t9.log 2>&1
echo "pytest(deep) rc $?"; tail -3 gpurun_out/t9.log
for d in 0 1; do
  echo "== deep $d"
  CPD_GC_ROWWAVE_DEEP=$d FRAMES=48 timeout 600 python tools/conv_bench.py sparse f16x2 10 2>&1 | grep -v amdgpu.ids
done
for d in 0 1 0 1; do
  CPD_GC_ROWWAVE_DEEP=$d timeout 600 python bench.py --streams 1 --no-cpu-baseline --no-extras > gpurun_out/bench_deep$d.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_deep$d.json'))
print("deep $d:", round(d["value"],1), {k: round(v['ms_per_frame'],4) for k,v in d["roofline"]["all_conv_kernels"].items() if 'rowwave' in k})
PY
done
