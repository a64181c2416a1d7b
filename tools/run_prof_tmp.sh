for f in 48; do for lib in tools/probe/libcpd_head.so cpd_amd/csrc/libcpd_hip.so; do echo "FRAMES=$f $lib"; CPD_HIP_LIB=$PWD/$lib FRAMES=$f python tools/conv_bench.py dense f16x2 20 2>&1 | grep -v amdgpu.ids | grep "tile_conv"; done; done
python -m pytest tests/test_gpu_dense.py tests/test_gpu_train.py tests/test_gpu_autograd.py -x -q 2>&1 | tail -3
