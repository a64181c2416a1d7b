export CPD_HIP_LIB=$PWD/cpd_amd/csrc/libcpd_hip.so
for r in 1 2 3; do for d in _ab_prev .; do (cd $d && python bench.py --mode train --steps 300 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null) | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-10s %.3f ms/step' % (sys.argv[1], d['ms_per_step']))" $d; done; done
