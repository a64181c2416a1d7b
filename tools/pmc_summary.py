#!/usr/bin/env python3
"""Summarise the three rocprofv3 --pmc passes of tools/pmc_bench.sh per conv kernel.
FETCH_SIZE / WRITE_SIZE are in KB (MI355X_MICROARCH.md, HBM / rocprofv3 section); gfx950 reports half
of wide coalesced reads, so FETCH is doubled; WRITE_SIZE is taken as is."""
import collections
import csv
import glob
import json
import re
import sys

root, out = sys.argv[1], sys.argv[2]


def short(name):
    m = re.search(r"((?:gather|tile|rowwave|window)_conv\w*kernel)<([^>]*)>", name)
    if not m:
        return None
    args = [a.strip() for a in m.group(2).split(",")]
    if m.group(1) == "tile_conv_bf16_kernel":
        args = args[:2]
    if m.group(1).startswith("window_conv") and len(args) == 2 and args[1] == "128":
        args = args[:1]                      # default rows-per-workgroup template argument: the bench's name omits it
    return "%s<%s>" % (m.group(1), ",".join(args))


vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/*/*counter_collection.csv") + glob.glob(root + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, c in sorted(vals.items()):
    avg = {n: sum(v) / len(v) for n, v in c.items()}
    e = {"launches": len(c.get("FETCH_SIZE", [])), "FETCH_SIZE_KB_avg": round(avg.get("FETCH_SIZE", 0.0), 1),
         "WRITE_SIZE_KB_avg": round(avg.get("WRITE_SIZE", 0.0), 1)}
    e["hbm_bytes_per_launch_corrected"] = int(1024 * (2 * avg.get("FETCH_SIZE", 0.0) + avg.get("WRITE_SIZE", 0.0)))
    if "GRBM_GUI_ACTIVE" in avg and avg["GRBM_GUI_ACTIVE"] > 0:
        e["mfma_busy_frac_of_simd_cycles"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (avg["GRBM_GUI_ACTIVE"] / 8.0), 3)
    res[k] = e
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import bench
    src_hash = bench.kernel_source_hash()
except Exception:
    src_hash = None
json.dump({"git_sha": os.environ.get("CPD_GIT_SHA"), "kernel_source_hash": src_hash,
           "stamp_note": "git_sha = HEAD of the tree the passes ran on (CPD_GIT_SHA, set by the gpurun command line: the GPU box has no "
                         ".git); kernel_source_hash = bench.kernel_source_hash() over cpd_amd/csrc/*.hip|*.h of that tree -- bench.py "
                         "prints both next to roofline.traffic and says whether its own kernel sources hash the same",
           "source": "rocprofv3 --pmc passes of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline` "
                     "(FETCH_SIZE, WRITE_SIZE and SQ_VALU_MFMA_BUSY_CYCLES/GRBM_GUI_ACTIVE in three separate passes, "
                     "--kernel-trace only; tools/pmc_bench.sh). FETCH_SIZE is doubled (gfx950 reports 1/2 of wide coalesced "
                     "reads, MI355X_MICROARCH.md); WRITE_SIZE is uncalibrated. mfma busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                     "(1024 SIMDs x GRBM_GUI_ACTIVE/8).", "kernels": res}, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
