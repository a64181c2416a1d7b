#!/usr/bin/env python3
"""Is the dense window kernel power-limited? Runs the 128->128 3x3 layer of config 2 (16 frames) in a loop for a few seconds
per input distribution -- N(0,1), post-ReLU-like (half zeros), all zeros, constant -- with the SAME instruction stream, and
samples rocm-smi (clock, power) in the background. A large time difference between random and zero inputs at equal instruction
counts means the clock (power management), not the schedule, sets the time.
usage: python tools/power_probe.py [seconds per case]"""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cpd_amd import ops

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
B, hw, cin, cout = 16, 188, 128, 128
nbr, ho, wo = ops.rulebook_conv2d(B, hw, hw, 3, 3, 1, 1, "cuda")
n = B * hw * hw
w = torch.randn(9, cin, cout, device="cuda") * (2.0 / (9 * cin)) ** 0.5
pw = ops.pack_weight(w)
pwz = ops.pack_weight(torch.zeros_like(w))
samples = []
stop = False


def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(o)
            c = d[sorted(d)[0]]
            samples.append((time.time(), {k: v for k, v in c.items() if "sclk" in k.lower() or "power" in k.lower()}))
        except Exception as e:  # noqa
            samples.append((time.time(), {"err": str(e)[:80]}))
        time.sleep(0.25)


th = threading.Thread(target=sampler, daemon=True)
th.start()
cases = [("randn", torch.randn(n, cin, device="cuda"), pw), ("relu(randn)", torch.randn(n, cin, device="cuda").relu(), pw),
         ("zeros", torch.zeros(n, cin, device="cuda"), pw), ("ones", torch.ones(n, cin, device="cuda"), pw),
         ("randn x zero weights", torch.randn(n, cin, device="cuda"), pwz)]
want = os.environ.get("PP_CASES")
if want:
    cases = [c for c in cases if c[0] in want.split(",")]
for math in os.environ.get("PP_MATH", "f16x2,bf16x3").split(","):
    for name, x, pk in cases:
        out = ops.gather_conv(x, cin, pk, nbr, 9, n, cout, dense=True, math=math)
        torch.cuda.synchronize()
        t0 = time.time()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 0
        e0.record()
        while time.time() - t0 < secs:
            for _ in range(50):
                ops.gather_conv(x, cin, pk, nbr, 9, n, cout, dense=True, math=math, out=out)
            reps += 50
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        t1 = time.time()
        us = e0.elapsed_time(e1) / reps * 1e3
        mine = [s for (t, s) in samples if t0 + 0.5 <= t <= t1]
        last = mine[-1] if mine else {}
        smi = " ".join(str(v) for v in last.values())
        print("%-7s %-22s %8.1f us  %6.1f TF   smi: %s" % (math, name, us, 2.0 * 9 * n * cin * cout / us / 1e6, smi), flush=True)
stop = True
