#!/bin/bash
# same-box A/B of one bench.py flag: tools/ab_flag.sh <rounds> "<flag A>" "<flag B>" -- <common bench args...>; prints frames/s per run
rounds=$1; A=$2; B=$3; shift 4
for r in $(seq $rounds); do
  for f in "$A" "$B"; do
    python bench.py --no-extras --no-cpu-baseline --no-roofline --no-digest-check "$@" $f 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-24s %8.1f frames/s  %.3f ms/step  digest %s' % (sys.argv[1], d['value'], d['ms_per_step'], (d.get('results_digest') or {}).get('timed_steps',['-'])[0]))" "$f"
  done
done
