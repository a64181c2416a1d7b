#!/bin/bash
# round 3 GPU call 8: the WHOLE -m gpu suite, stream count sweep, probes stored as JSON
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/t8.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/t8.log
for s in 1 2 3; do
  timeout 600 python bench.py --streams $s --no-cpu-baseline --no-extras --no-roofline > gpurun_out/bench_streams$s.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/bench_streams$s.json')); print('streams $s', round(d['value'],1), round(d['ms_per_step'],2))"
done
timeout 300 python tools/power_probe.py 3 > gpurun_out/r03_power_probe.txt 2>&1; tail -8 gpurun_out/r03_power_probe.txt
timeout 600 python tools/c5_stress.py > gpurun_out/r03_c5_stress.json 2> gpurun_out/c5.err; tail -c 1500 gpurun_out/r03_c5_stress.json
