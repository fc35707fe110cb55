"""Isolated timings of the pre-split GEMM (ops.linear_planes) at the training step's frame-sized shapes; FCL_PGEMM_CFG picks the tile configuration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fcl_taco2_amd import ops

dev = torch.device("cuda:0")
for m, n, k in ((24300, 1024, 256), (24300, 256, 1024), (24300, 512, 128), (24300, 128, 512), (12400, 4096, 512), (2480, 4096, 1024), (31000, 512, 512)):
    x = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) / k ** 0.5
    xp, wp = ops.pack_planes(x), ops.pack_planes(w)
    y = ops.linear_planes(xp, wp, n, k)[0]
    ref = x.double() @ w.double().t()
    err = float((y.double() - ref).abs().max())
    for _ in range(3):
        ops.linear_planes(xp, wp, n, k)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.linear_planes(xp, wp, n, k)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print("cfg %s  M %6d N %5d K %5d: %7.1f us  %6.1f TFLOP/s fp32-eq  (max err %.1e)" % (os.environ.get("FCL_PGEMM_CFG", "auto"), m, n, k, us, 2.0 * m * n * k / us / 1e6, err))
