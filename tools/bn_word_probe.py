import os, sys, torch
sys.path.insert(0, os.getcwd())
from cpd_amd import train_ops as T
def timeit(fn, reps=50):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
n, c = 188*188, 128
x = torch.randn(n, c, device="cuda"); dy = torch.randn(n, c, device="cuda") * 1e-5
mean, invstd = x.mean(0), (x.var(0, unbiased=False) + 1e-3).rsqrt(); gamma = torch.rand(c, device="cuda") + 0.5
y = torch.relu((x - mean) * invstd * gamma)
am = torch.zeros(T.ABSMAX_WORDS, dtype=torch.int32, device="cuda")
print("no word         %.1f us" % timeit(lambda: T.bn_backward(dy, y, x, mean, invstd, gamma)))
print("word (no zero)  %.1f us" % timeit(lambda: T.bn_backward(dy, y, x, mean, invstd, gamma, dx_absmax=am)))
def z():
    am.zero_(); T.bn_backward(dy, y, x, mean, invstd, gamma, dx_absmax=am)
print("word + zero     %.1f us" % timeit(z))
print("zero only       %.1f us" % timeit(lambda: am.zero_()))
