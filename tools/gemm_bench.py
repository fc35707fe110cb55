"""Micro-benchmark of the fp32-MFMA GEMM / LSTM-step kernels (developer tool; run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fcl_taco2_amd  # noqa
from fcl_taco2_amd import ops


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


def main():
    shapes = [(2500, 1024, 512), (2500, 1024, 256), (3200, 1024, 256), (3200, 256, 1280), (25000, 128, 640), (25000, 128, 400), (25000, 80, 640),
              (8192, 1024, 512), (16384, 1024, 1024), (1024, 1024, 512), (512, 1024, 512), (2500, 256, 256), (2500, 80, 256)]
    for (m, n, k) in shapes:
        x = torch.randn(m, k, device="cuda")
        w = torch.randn(n, k, device="cuda")
        b = torch.randn(n, device="cuda")
        us = bench(lambda: ops.linear(x, w, b, 1))
        print("linear M=%6d N=%5d K=%5d : %8.1f us  %6.1f TF" % (m, n, k, us, 2.0 * m * n * k / us / 1e6))


if __name__ == "__main__":
    main()
