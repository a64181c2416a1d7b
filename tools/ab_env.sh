#!/bin/bash
# same-box A/B of environment switches: tools/ab_env.sh <rounds> "<VAR=..> [VAR=..]" "<...>" ... -- <bench args>; "-" = no variables
rounds=$1; shift
sets=()
while [ "$1" != "--" ]; do sets+=("$1"); shift; done
shift
for r in $(seq $rounds); do
  for e in "${sets[@]}"; do
    if [ "$e" = "-" ]; then v=""; else v="$e"; fi
    env $v python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-60s %8.2f %s  %.3f ms/step  loss %s' % (sys.argv[1], d['value'], d['unit'], d['ms_per_step'], d.get('final_loss', d.get('config',{}).get('final_loss','-'))))" "$e"
  done
done
