"""Isolated timings of the training step's reduction kernels (BatchNorm statistics, column sums, grad-norm) at the KD step's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fcl_taco2_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for m, c in ((3200, 256), (3200, 512), (31000, 128), (31000, 512), (31000, 80)):
    z = torch.randn(m, c, device=dev)
    y = torch.randn(m, c, device=dev)
    g, b = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    out = torch.zeros(c, device=dev)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    t1 = timeit(lambda: ops.bn_stats(z, 1e-5, 0.1, rm, rv))
    t2 = timeit(lambda: ops.colsum(z, out))
    t3 = timeit(lambda: ops.colsum(z, out, y, g, b, mode=3))
    mean, invstd = ops.bn_stats(z, 1e-5)
    ref_m, ref_v = z.double().mean(0), z.double().var(0, unbiased=False)
    err = max(float((mean.double() - ref_m).abs().max()), float((invstd.double() - 1.0 / torch.sqrt(ref_v + 1e-5)).abs().max()))
    print("M %6d C %4d: bn_stats %6.1f us  colsum %6.1f us  colsum(mode 3) %6.1f us   (%.0f MB; stats err %.1e)" % (m, c, t1, t2, t3, m * c * 4 / 1e6, err))
for n in (6_500_000, 29_000_000):
    x = torch.randn(n, device=dev)
    acc = torch.zeros(1, device=dev, dtype=torch.float64)
    t = timeit(lambda: ops.sumsq_accum(x, acc)) if hasattr(ops, "sumsq_accum") else float("nan")
    print("sumsq n %9d: %6.1f us" % (n, t))
