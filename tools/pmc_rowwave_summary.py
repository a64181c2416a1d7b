#!/usr/bin/env python3
"""Per-kernel summary of tools/pmc_rowwave.sh: one JSON record per conv kernel of the bench with what its waves wait for.
usage: pmc_rowwave_summary.py <pass dir> <out.json> [name filter regex]"""
import collections, csv, glob, json, re, sys

root, outp = sys.argv[1], sys.argv[2]
filt = re.compile(sys.argv[3] if len(sys.argv) > 3 else r"(rowwave|rowplan|window|tile|gather)_conv")


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n).replace(" ", "")


vals = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(root + "/*/*counter_collection.csv") + glob.glob(root + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if filt.search(k):
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(root + "/sq1/*kernel_trace.csv") + glob.glob(root + "/sq1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if filt.search(k):
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)

res = {}
for k in sorted(vals):
    a = {c: sum(v) / len(v) for c, v in vals[k].items()}
    g = lambda n: a.get(n, 0.0)
    d = sum(dur[k]) / max(1, len(dur[k]))
    cyc = g("GRBM_GUI_ACTIVE") / 8.0                     # shader cycles of the launch (per XCD)
    wc = max(g("SQ_WAVE_CYCLES"), 1.0)
    waves = max(g("SQ_WAVES"), 1.0)
    e = {"launches": len(dur[k]), "dur_us": round(d, 1), "clk_ghz": round(cyc / d / 1e3, 3) if d else None,
         "waves": int(waves),
         # SQ_LEVEL_WAVES accumulates resident waves per cycle (per SE sample): mean resident waves per CU
         "occupancy_waves_per_cu_by_wave_cycles": round(4.0 * wc / max(cyc, 1) / 256.0, 2),
         "mfma_busy_frac": round(g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0 / max(cyc, 1), 3),
         "wave_cycles_frac": {"wait_any(s_waitcnt/barrier)": round(g("SQ_WAIT_ANY") / wc, 3),
                              "wait_inst_any(issue stall)": round(g("SQ_WAIT_INST_ANY") / wc, 3),
                              "active_inst_any": round(g("SQ_ACTIVE_INST_ANY") / wc, 3),
                              "active_vmem": round(g("SQ_ACTIVE_INST_VMEM") / wc, 3), "active_lds": round(g("SQ_ACTIVE_INST_LDS") / wc, 3),
                              "active_valu": round(g("SQ_ACTIVE_INST_VALU") / wc, 3), "active_scalar": round(g("SQ_ACTIVE_INST_SCA") / wc, 3),
                              "wait_inst_lds": round(g("SQ_WAIT_INST_LDS") / wc, 3)},
         "insts_per_wave": {n[9:].lower(): round(g(n) / waves, 1) for n in ("SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SALU",
                                                                           "SQ_INSTS_SMEM", "SQ_INSTS_VALU", "SQ_INSTS_MFMA")},
         "vmem_in_flight_per_cu(SQ_INST_LEVEL_VMEM/cycles/256)": round(g("SQ_INST_LEVEL_VMEM") / max(cyc, 1) / 256.0, 2),
         "tcp": {"cache_accesses": g("TCP_TOTAL_CACHE_ACCESSES_sum"), "tcc_read_req": g("TCP_TCC_READ_REQ_sum"),
                 "l1_hit_rate": round(1.0 - g("TCP_TCC_READ_REQ_sum") / max(g("TCP_TOTAL_CACHE_ACCESSES_sum"), 1.0), 3),
                 "tcc_read_latency_cycles": round(g("TCP_TCC_READ_REQ_LATENCY_sum") / max(g("TCP_TCC_READ_REQ_sum"), 1.0), 1),
                 "tcp_latency_cycles_per_access": round(g("TCP_TCP_LATENCY_sum") / max(g("TCP_TOTAL_CACHE_ACCESSES_sum"), 1.0), 1),
                 "busy_frac(GATE_EN2/GATE_EN1)": round(g("TCP_GATE_EN2_sum") / max(g("TCP_GATE_EN1_sum"), 1.0), 3),
                 "pending_stall_frac": round(g("TCP_PENDING_STALL_CYCLES_sum") / max(g("TCP_GATE_EN1_sum"), 1.0), 3),
                 "tcr_stall_frac": round(g("TCP_TCR_TCP_STALL_CYCLES_sum") / max(g("TCP_GATE_EN1_sum"), 1.0), 3),
                 "tagconflict_stall_frac": round(g("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum") / max(g("TCP_GATE_EN1_sum"), 1.0), 3),
                 "ta_data_stall_frac": round(g("TCP_TCP_TA_DATA_STALL_CYCLES_sum") / max(g("TCP_GATE_EN1_sum"), 1.0), 3)},
         "ta": {"busy_frac(TA_TA_BUSY/256cu/cycles)": round(g("TA_TA_BUSY_sum") / 256.0 / max(cyc, 1), 3), "busy_avr": g("TA_BUSY_avr"),
                "addr_stalled_by_tc_frac": round(g("TA_ADDR_STALLED_BY_TC_CYCLES_sum") / 256.0 / max(cyc, 1), 3),
                "data_stalled_by_tc_frac": round(g("TA_DATA_STALLED_BY_TC_CYCLES_sum") / 256.0 / max(cyc, 1), 3),
                "flat_read_wavefronts": g("TA_FLAT_READ_WAVEFRONTS_sum")},
         "td": {"busy_frac": round(g("TD_TD_BUSY_sum") / 256.0 / max(cyc, 1), 3), "tc_stall_frac": round(g("TD_TC_STALL_sum") / 256.0 / max(cyc, 1), 3)},
         "tcc": {"hit": g("TCC_HIT_sum"), "miss": g("TCC_MISS_sum"), "req": g("TCC_REQ_sum"),
                 "hit_rate": round(g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1.0), 3)},
         "hbm": {"FETCH_SIZE_KB": round(g("FETCH_SIZE"), 1), "WRITE_SIZE_KB": round(g("WRITE_SIZE"), 1),
                 "bytes_per_launch_corrected": int(1024 * (2 * g("FETCH_SIZE") + g("WRITE_SIZE"))),
                 "TBps": round(1024 * (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) / max(d, 1e-9) / 1e6, 3)},
         "derived": {"MemUnitStalled": g("MemUnitStalled"), "OccupancyPercent": g("OccupancyPercent")},
         "raw": {c: v for c, v in sorted(a.items())}}
    res[k] = e
json.dump({"source": "tools/pmc_rowwave.sh: separate rocprofv3 --kernel-trace --pmc passes of `python bench.py --steps 2 --warmup 1 "
                     "--no-cpu-baseline --no-roofline`; averages per launch. SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles "
                     "(MI355X_MICROARCH.md); FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads).", "kernels": res},
          open(outp, "w"), indent=1)
for k, e in res.items():
    if "rowwave" in k or "window_conv_f16_kernel<128,128>" in k:
        print(k, json.dumps({x: e[x] for x in ("dur_us", "clk_ghz", "occupancy_waves_per_cu_by_wave_cycles", "mfma_busy_frac", "wave_cycles_frac",
                                               "insts_per_wave", "tcp", "ta", "td", "tcc", "hbm", "derived")}))
