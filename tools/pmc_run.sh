#!/bin/bash
# usage: pmc_run.sh "<pmc_conv args>" tag
export TMPDIR=/tmp
args="$1"; tag="$2"; out=gpurun_out/pmc2/$tag; mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAIT_INST_LDS --output-format csv -d $out/a -o p -- python tools/pmc_conv.py $args >/dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD --output-format csv -d $out/b -o p -- python tools/pmc_conv.py $args >/dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/c -o p -- python tools/pmc_conv.py $args >/dev/null 2>&1
python - <<PY
import csv, collections, glob
agg=collections.defaultdict(list); dur=[]
for f in glob.glob("$out/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "conv_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for r in csv.DictReader(open("$out/a/p_kernel_trace.csv")):
    if "conv_kernel" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
a={k:sum(v)/len(v) for k,v in agg.items()}
cyc=a["GRBM_GUI_ACTIVE"]/8
print("$tag", "dur_us %.1f"%(sum(dur)/len(dur)), "clk GHz %.2f"%(cyc/(sum(dur)/len(dur))/1e3), "mfma_util %.3f"%(a["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/cyc),
  "wave: wait_any %.2f wait_inst %.2f active %.2f lds_wait %.3f"%(a["SQ_WAIT_ANY"]/a["SQ_WAVE_CYCLES"], a["SQ_WAIT_INST_ANY"]/a["SQ_WAVE_CYCLES"], a["SQ_ACTIVE_INST_ANY"]/a["SQ_WAVE_CYCLES"], a["SQ_WAIT_INST_LDS"]/a["SQ_WAVE_CYCLES"]),
  "lds_conflict/active %.3f"%(a["SQ_LDS_BANK_CONFLICT"]/max(1,a["SQ_LDS_IDX_ACTIVE"])), "waves %d"%a["SQ_WAVES"], "fetchKB %.0f"%a["FETCH_SIZE"])
PY
