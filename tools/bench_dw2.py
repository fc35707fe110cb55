"""Weight-gradient GEMM shapes of the KD / teacher updates through fcl_gemm_tn_taps_fwd (developer tool).
    python tools/bench_dw2.py            # one table per environment in CONFIGS (child processes: the tunables are read once per process)
    python tools/bench_dw2.py --child    # this process's configuration only"""
import os
import subprocess
import sys
import time

SHAPES = [  # (m, n, k, taps, what)
    (24300, 1024, 256, 1, "dec cells dg x h (x4), KD lstm taps (x2)"),
    (24300, 256, 256, 1, "prenet L1, KD prenet tap"),
    (24300, 256, 80, 1, "prenet L0"),
    (24300, 80, 256, 1, "feat_out"),
    (27500, 512, 128, 1, "KD postnet taps (x4)"),
    (27500, 128, 128, 5, "postnet conv (x3)"),
    (27500, 128, 80, 5, "postnet conv first"),
    (27500, 80, 128, 5, "postnet conv last"),
    (2700, 256, 256, 5, "encoder conv (x3)"),
    (2700, 384, 256, 3, "predictor conv 0 (dur)"),
    (2700, 384, 384, 3, "predictor conv 1 (dur)"),
    (2700, 256, 256, 3, "predictor conv (pitch / energy)"),
    (2700, 512, 256, 1, "BiLSTM input, KD enc taps"),
    (2700, 512, 128, 1, "BiLSTM recurrent"),
    (2500, 1024, 256, 1, "dG0 x att_c"),
    (12500, 4096, 1024, 1, "T: dec cells"),
    (12500, 4096, 256, 1, "T: dec cells x prenet"),
    (13800, 512, 512, 5, "T: postnet conv"),
    (1400, 512, 512, 5, "T: encoder conv"),
]
CONFIGS = [("old gemm_tn_kernel", {"FCL_DW_MFMA": "0"}), ("dw_mfma default", {}), ("dw_mfma all 128x128", {"FCL_DW_BIG_TILES_MIN": "1"}),
           ("dw_mfma all 64x64", {"FCL_DW_BIG_TILES_MIN": "100000"})]


def child():
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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

    tot = 0.0
    for m, n, k, taps, what in SHAPES:
        a, b = torch.randn(m, n, device=dev), torch.randn(m, k, device=dev)
        lo = torch.zeros(m, dtype=torch.int32, device=dev)
        hi = torch.full((m,), m, dtype=torch.int32, device=dev)
        if taps == 1:
            out = torch.zeros(n, k, device=dev)
            t = timeit(lambda: ops.gemm_tn(a, b, out))
        else:
            out = torch.zeros(taps, n, k, device=dev)
            t = timeit(lambda: ops.gemm_tn_taps(a, b, out, -(taps // 2), seg_lo=lo, seg_hi=hi))
        gf = 2.0 * m * n * k * taps / 1e6
        mb = 4.0 * m * (n + k) / 1e6
        tot += t
        print("  m %6d n %5d k %5d taps %d: %7.1f us %6.1f TF  (inputs %5.1f MB = %5.1f us at 4 TB/s)  %s" % (m, n, k, taps, t, gf / t, mb, mb / 4.0, what), flush=True)
    print("  sum %.1f us" % tot)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        for name, env in CONFIGS:
            print(name, env, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, **env), check=False)
