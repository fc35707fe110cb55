"""Isolated Conv1d-on-planes timings (developer tool): FCL_PCONV=0/1 python tools/bench_conv.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import fcl_taco2_amd
from fcl_taco2_amd import ops
from fcl_taco2_amd.plan import ConvPack
dev = "cuda:0"
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
class CV: pass
for m, cin, cout, k in [(3200, 256, 256, 5), (3200, 256, 384, 3), (25026, 128, 128, 5), (25026, 80, 128, 5), (1600, 512, 512, 5), (13800, 512, 512, 5)]:
    x = torch.randn(m, cin, device=dev)
    w = torch.randn(cout, cin, k, device=dev) * 0.05
    wp = ops.pack_conv1d_weight(w)
    cv = CV(); cv.wpp = ops.pack_planes(wp.reshape(k * cout, cin)); cv.bias = None; cv.cout, cv.cin, cv.k = cout, cin, k
    lo = torch.zeros(m, dtype=torch.int32, device=dev); hi = torch.full((m,), m, dtype=torch.int32, device=dev)
    xp = ops.pack_planes(x)
    us = t(lambda: ops.conv1d_planes(xp, cv, lo, hi, ops.ACT_NONE, want_f32=False, want_planes=True))
    print("m %6d cin %4d cout %4d k %d: %7.1f us  %6.1f TF" % (m, cin, cout, k, us, 2.0 * m * cin * cout * k / us / 1e6))
