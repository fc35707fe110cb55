import os, sys
sys.path.insert(0, "/root/repo")
import torch
from fcl_taco2_amd import ops
dev = torch.device("cuda:0")
for kind in ("zeros",):
    for m, n, k in ((24320, 1024, 256), (24320, 512, 128)):
        mk = (lambda *s: torch.randn(*s, device=dev)) if kind == "randn" else ((lambda *s: torch.zeros(*s, device=dev)) if kind == "zeros" else (lambda *s: torch.ones(*s, device=dev)))
        x, w = mk(m, k), mk(n, k)
        xp, wp = ops.pack_planes(x), ops.pack_planes(w)
        y = torch.empty(m, n, device=dev)
        lib = ops._lib.load()
        def call():
            ops.check(lib.fcl_linear_planes_fwd(xp.data_ptr(), xp.shape[1] // 64, wp.data_ptr(), None, y.data_ptr(), n, None, m, n, k, 0, torch.cuda.current_stream().cuda_stream))
        for _ in range(3): call()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): call()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1e3
        print("%-6s M %6d N %5d K %5d: %7.1f us  %6.1f TFLOP/s" % (kind, m, n, k, us, 2.0 * m * n * k / us / 1e6))
