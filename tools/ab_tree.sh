#!/bin/bash
# same-box A/B of two source trees sharing the in-tree library: tools/ab_tree.sh <rounds> <dir A> <dir B> -- <bench args>   (a tree = a
# directory holding bench.py + cpd_amd/, e.g. `git archive <rev> cpd_amd bench.py | tar -x -C _ab_prev`)
rounds=$1; A=$2; B=$3; shift 4
export CPD_HIP_LIB=$PWD/cpd_amd/csrc/libcpd_hip.so
for r in $(seq $rounds); do
  for d in "$A" "$B"; do
    (cd $d && python bench.py --no-extras --no-cpu-baseline --no-roofline --no-digest-check "$@" 2>/dev/null) | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-12s %8.1f frames/s  %.3f ms/step  digest %s' % (sys.argv[1], d['value'], d['ms_per_step'], (d.get('results_digest') or {}).get('timed_steps',['-'])[0]))" "$d"
  done
done
