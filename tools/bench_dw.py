"""Weight-gradient GEMM: fcl_gemm_tn_fwd (fp32 operands, in-kernel split) vs fcl_pack_planes_t x 2 + fcl_gemm_tn_planes (developer tool)."""
import sys
import time

import torch

sys.path.insert(0, ".")
import fcl_taco2_amd  # noqa
from fcl_taco2_amd import ops

dev = "cuda:0"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for m, n, k in [(12500, 4096, 1024), (12500, 4096, 256), (13800, 512, 512), (25000, 1024, 256), (25000, 256, 256), (3200, 512, 512), (25000, 80, 512)]:
    a, b = torch.randn(m, n, device=dev), torch.randn(m, k, device=dev)
    out = torch.zeros(n, k, device=dev)
    t_old = timeit(lambda: ops.gemm_tn(a, b, out))
    t_pa = timeit(lambda: ops.pack_planes_t(a))
    t_pb = timeit(lambda: ops.pack_planes_t(b))
    ap, bp = ops.pack_planes_t(a), ops.pack_planes_t(b)
    t_g = timeit(lambda: ops.gemm_tn_planes(ap, bp, out, m))
    with ops.gemm_mode("bf16"):
        t_gb = timeit(lambda: ops.gemm_tn_planes(ap, bp, out, m))
    gf = 2.0 * m * n * k / 1e6
    print("m %6d n %5d k %5d: gemm_tn %7.1f us (%5.1f TF) | pack_t %6.1f + %6.1f, gemm_tn_planes %7.1f us (%5.1f TF), bf16 %7.1f us" % (
        m, n, k, t_old, gf / t_old, t_pa, t_pb, t_g, gf / t_g, t_gb))
