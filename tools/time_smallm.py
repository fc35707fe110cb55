"""Y = A W^T on a few rows (the per-step GEMMs of the BPTT tails, gemm_smallm_kernel): time and exactness per tile rule (developer aid).
Usage: python tools/time_smallm.py   (runs itself with FCL_GEMM_SMALLM_32_MIN_WG=1000000000 = the 16 x 16 tiles of round 4, then the default)"""
import os, subprocess, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import fcl_taco2_amd  # noqa
    from fcl_taco2_amd import ops
    g = torch.Generator().manual_seed(0)
    for (N, K) in ((2048, 4096), (512, 1024), (2048, 1024)):
        w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        for M in (8, 16, 48, 64, 100, 128, 200, 256):
            x = torch.randn(M, K, generator=g).cuda()
            y = ops.linear(x, w)
            ref = (x.double() @ w.double().t()).float()
            err = float((y - ref).abs().max())
            for _ in range(5):
                ops.linear(x, w)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                ops.linear(x, w)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 50 * 1e3
            print("M=%3d N=%4d K=%4d  %6.1f us  %6.1f TFLOP/s  max err %.1e" % (M, N, K, us, 2.0 * M * N * K / us / 1e6, err))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run()
    else:
        for v in ("1000000000", None):
            env = dict(os.environ)
            if v:
                env["FCL_GEMM_SMALLM_32_MIN_WG"] = v
            print("FCL_GEMM_SMALLM_32_MIN_WG=%s" % (v or "default"))
            sys.stdout.flush()
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
