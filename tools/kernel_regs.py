#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_regs.py cpd_amd/csrc/gather_conv.hip [name-substring] [-- extra hipcc flags]"""
import re, subprocess, sys
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "--" else ""
extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True).stderr
cur = None
rows = []
for ln in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs Spill|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", ln)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        cur = {"name": name}
        rows.append(cur)
    else:
        cur[{"VGPRs Spill": "spill"}.get(k, k.split(" ")[0])] = v
print("%-62s %5s %5s %7s %5s %6s %7s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "spill", "LDS"))
for r in rows:
    if pat in r["name"]:
        print("%-62s %5s %5s %7s %5s %6s %7s" % (r["name"][:62], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("Occupancy"),
                                                   r.get("spill"), r.get("LDS")))
