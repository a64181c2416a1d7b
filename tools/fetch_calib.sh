#!/bin/bash
# FETCH_SIZE calibration on known gathers (tools/fetch_calib.hip): prints per kernel the counter (KB) next to the bytes really read
R=$PWD; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/fetch_calib
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/fetch_calib -o p -- $R/tools/probe/fetch_calib > $R/gpurun_out/fetch_calib.txt 2>/dev/null
rm -rf $R/gpurun_out/fetch_calib_w
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $R/gpurun_out/fetch_calib_w -o p -- $R/tools/probe/fetch_calib > /dev/null 2>&1
cd $R
cat gpurun_out/fetch_calib.txt
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/fetch_calib", "gpurun_out/fetch_calib_w"):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no counter file"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    known = {"k_stream": 2**31, "k_rows8": 2**31 + 2**26, "k_quad2": 2**31 + 2**26, "k_half64": 2**30 + 2**26}
    for (k, c), v in sorted(acc.items()):
        a = sum(v) / len(v)
        kb = known.get(k)
        if c == "FETCH_SIZE" and kb:
            print("%-10s %-22s avg %14.1f KB = %6.3f of the known bytes (x2: %5.3f)" % (k, c, a, a * 1024 / kb, 2 * a * 1024 / kb))
        else:
            print("%-10s %-22s avg %14.1f" % (k, c, a))
PY
