#!/usr/bin/env python3
"""Per-kernel summary of the tools/pmc_any.sh passes. usage: pmc_any_summary.py <dir> [name filter]"""
import collections, csv, glob, json, re, sys
out, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)


agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if filt in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/a/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if filt in k:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {}
for k in sorted(agg):
    a = {c: sum(v) / len(v) for c, v in agg[k].items()}
    d = sum(dur[k]) / max(1, len(dur[k]))
    cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8              # per XCD
    wc = max(a.get("SQ_WAVE_CYCLES", 1), 1)
    row = dict(launches=len(dur[k]), dur_us=d, clk_ghz=cyc / d / 1e3 if d else 0,
               mfma_busy=a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc if cyc else 0,
               wait_any=a.get("SQ_WAIT_ANY", 0) / wc, wait_inst=a.get("SQ_WAIT_INST_ANY", 0) / wc, active=a.get("SQ_ACTIVE_INST_ANY", 0) / wc,
               lds_wait=a.get("SQ_WAIT_INST_LDS", 0) / wc, lds_conflict=a.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, a.get("SQ_LDS_IDX_ACTIVE", 1)),
               waves=a.get("SQ_WAVES", 0), insts_mfma=a.get("SQ_INSTS_MFMA", 0), insts_valu=a.get("SQ_INSTS_VALU", 0),
               insts_lds=a.get("SQ_INSTS_LDS", 0), insts_vmem_rd=a.get("SQ_INSTS_VMEM_RD", 0),
               fetch_MB_x2=2 * a.get("FETCH_SIZE", 0) / 1e3, write_MB=a.get("WRITE_SIZE", 0) / 1e3)
    res[k] = row
    print("%-40s n=%3d %8.1f us clk %.2f mfma_busy %.3f | wave: wait %.2f stall %.2f active %.2f lds_stall %.3f | lds_confl %.3f | "
          "per wave: mfma %.0f valu %.0f lds %.0f vmem_rd %.0f | fetch(x2) %.0f MB write %.0f MB" %
          (k[:40], row["launches"], d, row["clk_ghz"], row["mfma_busy"], row["wait_any"], row["wait_inst"], row["active"], row["lds_wait"],
           row["lds_conflict"], row["insts_mfma"] / max(1, row["waves"]), row["insts_valu"] / max(1, row["waves"]),
           row["insts_lds"] / max(1, row["waves"]), row["insts_vmem_rd"] / max(1, row["waves"]), row["fetch_MB_x2"], row["write_MB"]))
json.dump(res, open(out + "/summary.json", "w"), indent=1)
