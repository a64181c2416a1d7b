import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpd_amd import ops
batch, hw, cin, cout, ms, nt, abl = [int(v) for v in sys.argv[1:8]]
os.environ["CPD_GC_MS"], os.environ["CPD_GC_NT"] = str(ms), str(nt)
nbr, ho, wo = ops.rulebook_conv2d(batch, hw, hw, 3, 3, 1, 1, "cuda")
n = batch * hw * hw
x = torch.randn((n, cin), device="cuda")
packed = ops.pack_weight(torch.randn((9, cin, cout), device="cuda") * 0.05)
out = torch.empty((n, cout), device="cuda")
for _ in range(2):
    ops.gather_conv(x, cin, packed, nbr, 9, n, cout, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.gather_conv(x, cin, packed, nbr, 9, n, cout, out=out)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10
flops = 2.0 * float((nbr >= 0).sum()) * cin * cout
print("batch %d %dx%d %d->%d tile(%d,%d) ablate=%d : %.1f us  %.1f TF" % (batch, hw, hw, cin, cout, ms, nt, abl, t * 1e3, flops / t / 1e9), flush=True)
