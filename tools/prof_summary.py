#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats csv next to the bench JSON of the same run."""
import csv
import json
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)


stats, bench, steps = sys.argv[1], sys.argv[2], sys.argv[3]
b = json.load(open(bench))
print("fps %.1f  ms/step %.2f" % (b["value"], b["ms_per_step"]))
rows = list(csv.DictReader(open(stats)))
if steps == "auto":      # steps executed under the profiler (warm-up, timed and roofline passes): one decode launch per step
    steps = float(sum(int(r["Calls"]) for r in rows if "decode_kernel" in r["Name"]) or 1)
else:
    steps = float(steps)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
conv = sum(float(r["TotalDurationNs"]) for r in rows if "conv_kernel" in r["Name"])
print("kernel ms/step: total %.2f  conv %.2f  other %.2f" % (tot / 1e6 / steps, conv / 1e6 / steps, (tot - conv) / 1e6 / steps))
for r in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 30]:
    print("%-58s calls/step %6.1f  avg us %8.1f  ms/step %6.3f" % (short(r["Name"])[:58], int(r["Calls"]) / steps,
                                                                  float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6 / steps))
