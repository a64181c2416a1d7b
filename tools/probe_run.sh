#!/bin/bash
# ablation probe: time the same conv with B loads / A gathers removed (diagnostic builds)
for cfg in "4 188 128 128 2 8" "1 188 128 128 2 2" "4 188 128 128 4 4" "4 188 128 128 2 4"; do
  for a in 0 1 2 3; do
    if [ $a = 0 ]; then unset CPD_HIP_LIB; else export CPD_HIP_LIB=$PWD/tools/probe/libcpd_abl$a.so; fi
    python tools/probe_time.py $cfg $a
  done
done
