"""Ablation timing of rowwave_conv_bf16_kernel on a dense 128->128 3x3 layer (diagnostic libs in tools/probe)."""
import os, sys, torch
os.environ["CPD_TUNE"] = "1"      # CPD_GC_* knobs are only read with this set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CPD_GC_DENSE_ROWWAVE"] = "1"
from cpd_amd import ops
batch, hw, cin, cout = 8, 188, int(sys.argv[1]), int(sys.argv[2])
nbr, ho, wo = ops.rulebook_conv2d(batch, hw, hw, 3, 3, 1, 1, "cuda")
n = batch * hw * hw
x = torch.randn((n, cin), device="cuda")
packed = ops.pack_weight(torch.randn((9, cin, cout), device="cuda") * 0.05)
out = torch.empty((n, cout), device="cuda")
for _ in range(2):
    ops.gather_conv(x, cin, packed, nbr, 9, n, cout, out=out, dense=True, bf16x3=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.gather_conv(x, cin, packed, nbr, 9, n, cout, out=out, dense=True, bf16x3=True)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10
print("%s %d->%d: %.1f us  %.1f TF-equivalent" % (os.environ.get("CPD_HIP_LIB", "full")[-12:], cin, cout, t * 1e3, 2.0 * n * 9 * cin * cout / t / 1e9), flush=True)
