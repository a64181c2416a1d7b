#!/bin/bash
# usage: tools/local_pmc.sh   (GPU box, repo root): counter passes (kernel-trace + one block each, as the pool requires) of the row-wave
# kernel on the real and on the perfect-locality rulebook of a level (tools/local_probe_one.py) -> gpurun_out/local_pmc.txt
export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/local_pmc; rm -rf $out; mkdir -p $out
cd /tmp
for lvl in ${LEVELS:-1 3}; do for v in real local; do
  i=0
  for pass in "TD_TD_BUSY_sum TD_TC_STALL_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE"; do
    i=$((i+1))
    LEVEL=$lvl VARIANT=$v timeout -k 5 240 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/l${lvl}_$v/$i -o p -- python $R/tools/local_probe_one.py > $out/l${lvl}_${v}_$i.log 2>&1 || echo "pass failed: $lvl $v $pass" >&2
  done
done; done
cd $R
python - <<'PY' | tee gpurun_out/local_pmc.txt
import csv, glob, collections, re
for d in sorted(glob.glob("gpurun_out/local_pmc/l*_*")):
    if not d.split("/")[-1].startswith("l") or "." in d.split("/")[-1]: continue
    agg = collections.defaultdict(list); dur = []
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "rowwave_conv" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/1/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "rowwave_conv" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    a = {k: sum(v) / len(v) for k, v in agg.items()}
    if not dur: continue
    us = sum(dur[1:]) / max(1, len(dur) - 1)
    cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8
    print("%-10s %8.1f us | TD busy %.2f  TA busy %.2f | L2 hit %.3f (req %.1f M) | L1 accesses %.1f M, L1->L2 reads %.1f M | MFMA busy %.3f | fabric fetch x2 %.0f MB" % (
        d.split("/")[-1], us, a.get("TD_TD_BUSY_sum", 0) / 256 / max(cyc, 1), a.get("TA_TA_BUSY_sum", 0) / 256 / max(cyc, 1),
        a.get("TCC_HIT_sum", 0) / max(1, a.get("TCC_HIT_sum", 0) + a.get("TCC_MISS_sum", 0)), a.get("TCC_REQ_sum", 0) / 1e6,
        a.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / 1e6, a.get("TCP_TCC_READ_REQ_sum", 0) / 1e6,
        a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / max(cyc, 1), 2 * a.get("FETCH_SIZE", 0) / 1e3))
PY
