#!/usr/bin/env python3
"""Timeline reading of a rocprofv3 --kernel-trace csv: per HIP queue, the busy time, the idle gaps between consecutive kernels and the
kernels that follow the largest gaps -- is a step bound by kernel time or by the space between launches?
usage: tools/trace_gaps.py <kernel_trace.csv> [skip_fraction=0.3]   (the first fraction of the trace -- warm-up -- is skipped)
       tools/trace_gaps.py <kernel_trace.csv> --between <kernel name substring> <i> <j>   (the window from the end of that kernel's i-th
       launch to the end of its j-th: e.g. adam_kernel 20 30 = ten train steps)"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if len(sys.argv) > 2 and sys.argv[2] == "--between":
    marks = [int(r["End_Timestamp"]) for r in rows if sys.argv[3] in r["Kernel_Name"]]
    lo, hi = marks[int(sys.argv[4])], marks[int(sys.argv[5])]
    rows = [r for r in rows if lo < int(r["Start_Timestamp"]) and int(r["End_Timestamp"]) <= hi]
    print("window: %d launches of %s" % (int(sys.argv[5]) - int(sys.argv[4]), sys.argv[3]))
else:
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
    cut = t0 + (t1 - t0) * skip
    rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
span = (max(int(r["End_Timestamp"]) for r in rows) - int(rows[0]["Start_Timestamp"])) / 1e6
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:60]
byq = collections.defaultdict(list)
for r in rows:
    byq[(r["Queue_Id"], r["Stream_Id"])].append(r)
print("span %.2f ms, %d kernels" % (span, len(rows)))
# union busy time over all queues
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
busy, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print("GPU busy (any queue) %.2f ms = %.1f %% of the span" % (busy / 1e6, busy / 1e4 / span))
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    kt = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e6
    gaps = collections.Counter(); gapn = collections.Counter(); tot_gap = 0; small = 0
    for a, b in zip(rs, rs[1:]):
        g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
        if g > 0:
            tot_gap += g
            if g < 20000: small += g
            gaps[short(b["Kernel_Name"])] += g; gapn[short(b["Kernel_Name"])] += 1
    print("queue %s stream %s: %d kernels, kernel time %.2f ms, gaps %.2f ms (of which gaps < 20 us: %.2f ms)" % (q[0], q[1], len(rs), kt, tot_gap / 1e6, small / 1e6))
    for n, g in gaps.most_common(12):
        print("    %-60s gap before it: total %.3f ms over %d (avg %.1f us)" % (n, g / 1e6, gapn[n], g / 1e3 / gapn[n]))
    kt_by = collections.Counter(); kn_by = collections.Counter()
    for r in rs:
        kt_by[short(r["Kernel_Name"])] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); kn_by[short(r["Kernel_Name"])] += 1
    print("    -- kernel time on this queue")
    for n, t in kt_by.most_common(22):
        print("    %-60s %8.3f ms over %5d launches (avg %.1f us)" % (n, t / 1e6, kn_by[n], t / 1e3 / kn_by[n]))
