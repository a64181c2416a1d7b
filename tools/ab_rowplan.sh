#!/bin/bash
# same-box A/B of the sparse row kernels: bench.py (no extras) per row order, printing the row-kernel lines.  tools/ab_rowplan.sh taps bricks ...
mkdir -p gpurun_out
for o in "$@"; do
  extra=""; ro=$o
  if [ "$o" = "bricks+plan" ]; then ro=bricks; extra="--plan 1 --plan-tile 256"; fi
  if [ "$o" = "bricks+plan128" ]; then ro=bricks; extra="--plan 1 --plan-tile 128"; fi
  timeout 300 python bench.py --no-extras --no-cpu-baseline --row-order $ro $extra > gpurun_out/ab_$o.json 2> gpurun_out/ab_$o.err || tail -5 gpurun_out/ab_$o.err
  python - <<PY
import json
d = json.load(open("gpurun_out/ab_$o.json")); r = d["roofline"]
print("$o", round(d["value"], 1), {k.replace("_kernel", "").replace("_conv", ""): round(v["ms_per_frame"], 4) for k, v in r["all_conv_kernels"].items() if "row" in k},
      {k: round(v["us_per_frame"], 1) for k, v in d["hbm_stages"].items() if isinstance(v, dict)})
PY
done
