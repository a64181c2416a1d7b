#!/bin/bash
# same-box A/B of library builds: tools/ab_lib.sh <rounds> <lib A | -> <lib B> ... ; "-" = the in-tree library. One stream, roofline pass on.
rounds=$1; shift
for r in $(seq $rounds); do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset CPD_HIP_LIB; else export CPD_HIP_LIB=$PWD/$lib; fi
    python bench.py --no-extras --no-cpu-baseline --streams 1 --no-digest-check > /tmp/ab_lib.json 2>/dev/null
    python - "$lib" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_lib.json"))
k = d["roofline"]["all_conv_kernels"]
rw = {n.split("<")[1].rstrip(">"): round(v["ms_per_frame"], 4) for n, v in k.items() if n.startswith("rowwave_conv_f16pe")}
print("%-28s %7.1f frames/s  row-wave %s  sum %.4f" % (sys.argv[1], d["value"], rw, sum(rw.values())))
PY
  done
done
