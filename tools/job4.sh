#!/bin/bash
# round 3 GPU call 4: ablation of the quad-gather row-wave kernel (48 frames, canonical order) + remaining new tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for a in 0 1 2 4 8 7 4096; do
  if [ $a = 0 ]; then unset CPD_HIP_LIB; else export CPD_HIP_LIB=$PWD/tools/probe/libcpd_abl$a.so; fi
  echo "== ablate $a" >> gpurun_out/abl_quad.txt
  FRAMES=48 timeout 300 python tools/conv_bench.py sparse f16x2 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/abl_quad.txt
done
unset CPD_HIP_LIB
cat gpurun_out/abl_quad.txt
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_train.py -x -q -m gpu -k "range_guard or fused_eval or full_size_config3" > gpurun_out/t4.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/t4.log
