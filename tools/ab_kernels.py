#!/usr/bin/env python3
"""Same-box A/B of two bench.py configurations, kernel by kernel: tools/ab_kernels.py "<args A>" "<args B>" [rounds]
Runs bench.py --no-extras --no-cpu-baseline with each argument string alternately and prints value + per-kernel ms/frame."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
a, b = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
res = {a: [], b: []}
for r in range(rounds):
    for cfg in (a, b):
        out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--no-extras", "--no-cpu-baseline"] + cfg.split(),
                             capture_output=True, text=True)
        try:
            res[cfg].append(json.loads(out.stdout.strip().splitlines()[-1]))
        except Exception:
            print("FAILED", cfg, out.stderr[-2000:])
            raise
for cfg in (a, b):
    print("==", cfg, " values:", [round(d["value"], 1) for d in res[cfg]])
keys = sorted(set(k for cfg in res for d in res[cfg] for k in d.get("roofline", {}).get("all_conv_kernels", {})))
if keys:
    print("%-44s %12s %12s" % ("kernel ms/frame (mean)", "A", "B"))
    tot = {a: 0.0, b: 0.0}
    for k in keys:
        row = []
        for cfg in (a, b):
            v = [d["roofline"]["all_conv_kernels"][k]["ms_per_frame"] for d in res[cfg] if k in d["roofline"]["all_conv_kernels"]]
            m = sum(v) / len(v) if v else 0.0
            tot[cfg] += m
            row.append(m)
        print("%-44s %12.4f %12.4f" % (k, row[0], row[1]))
    print("%-44s %12.4f %12.4f" % ("sum conv", tot[a], tot[b]))
    for st in ("voxelize+mean_vfe", "rulebook_subm", "rulebook_conv", "densify", "sparse_conv_c<=16"):
        row = []
        for cfg in (a, b):
            v = [d["hbm_stages"][st]["us_per_frame"] for d in res[cfg] if st in d.get("hbm_stages", {})]
            row.append(sum(v) / len(v) if v else 0.0)
        print("%-44s %12.2f %12.2f  us/frame" % (st, row[0], row[1]))
