#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_train.py -x -q -m gpu -k "range_guard or fused_eval or full_size_config3" > gpurun_out/t6.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/t6.log
# where does a 1-frame step go? kernel time (rocprofv3) vs wall, and the host side (cProfile)
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_1f -- python $R/bench.py --frames 1 --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/bench_1f_prof.json 2>/dev/null
cd $R
cp $(ls gpurun_out/prof_1f/*/*kernel_stats.csv | head -1) gpurun_out/r03_1frame_kernel_stats.csv
python - <<'PY'
import csv, json
rows = list(csv.DictReader(open('gpurun_out/r03_1frame_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
d = json.load(open('gpurun_out/bench_1f_prof.json'))
print("1 frame: wall %.3f ms/step; kernel time %.3f ms/step over %.1f launches/step" % (d['ms_per_step'], tot / 220 / 1e6, calls / 220))
PY
python - <<'PY' > gpurun_out/r03_1frame_cprofile.txt 2>&1
import cProfile, pstats, sys, torch
sys.path.insert(0, '.')
from cpd_amd.engine import CenterPointEngine, ModelConfig, init_state_dict
from cpd_amd.synthetic import waymo_cloud
cfg = ModelConfig(); sd = init_state_dict(cfg, 0)
eng = CenterPointEngine(cfg, sd, host_results=True)
pts = [torch.from_numpy(waymo_cloud(s)).cuda() for s in range(4)]
for i in range(10): eng.forward([pts[i % 4]])
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(100): eng.forward([pts[i % 4]])
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
PY
head -70 gpurun_out/r03_1frame_cprofile.txt
