python -m pytest tests/test_gpu_sparse.py tests/test_gpu_pipeline.py tests/test_gpu_train.py tests/test_gpu_train_ops.py tests/test_gpu_autograd.py tests/test_gpu_streams.py -x -q 2>&1 | tail -3
for lib in tools/probe/libcpd_head.so cpd_amd/csrc/libcpd_hip.so; do
CPD_HIP_LIB=$PWD/$lib python bench.py --no-extras --no-cpu-baseline --no-roofline --api modules 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib modules', d['value'], d['ms_per_step'])"
CPD_HIP_LIB=$PWD/$lib python bench.py --mode train --steps 40 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib train', d['value'], d['ms_per_step'])"
CPD_HIP_LIB=$PWD/$lib python bench.py --mode train --frames 8 --steps 12 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib train8', d['value'], d['ms_per_step'])"
done
python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', d['value'], d['ms_per_step'], d['roofline']['frac'], d['results_digest'].get('equal_to_single_stream_pass'))"
