#!/bin/bash
# same-box A/B of two library builds on the small-batch and train steps: tools/ab_lib_small.sh <rounds> <lib A> <lib B>   ("-" = in-tree)
rounds=$1; shift
for r in $(seq $rounds); do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset CPD_HIP_LIB; else export CPD_HIP_LIB=$PWD/$lib; fi
    for fr in 1 4; do
      python bench.py --no-extras --no-cpu-baseline --no-roofline --no-digest-check --frames $fr --streams 1 --steps 80 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-36s %d frame(s) %8.1f frames/s  %.3f ms/step  digest %s' % (sys.argv[1], d['config']['frames_per_step_per_gpu'], d['value'], d['ms_per_step'], d['results_digest']['timed_steps'][0]))" "$lib"
    done
    python bench.py --mode train --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-36s train %.3f ms/step  loss %s' % (sys.argv[1], d['ms_per_step'], d['config']['final_loss']))" "$lib"
  done
done
