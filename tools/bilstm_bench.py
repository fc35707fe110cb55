"""BiLSTM recurrence kernels in isolation (developer aid): forward (synthesis), training forward (saves gates) and BPTT at FCL-taco2-S size
(B = 32, T = 100, H = 128) or, with BILSTM_BENCH_MODEL=teacher, FCL-taco2-T size (B = 16, H = 256: the 4-workgroup kernels), 50 launches each under
HIP events; compares against the round-4 kernels (FCL_BILSTM_KSPLIT=0 FCL_BILSTM_GROUP_LL=0 in a child).
Usage: [BILSTM_BENCH_MODEL=teacher] python tools/bilstm_bench.py"""
import os, subprocess, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fcl_taco2_amd  # noqa
from fcl_taco2_amd import ops


def run():
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(1)
    B, T, C, H = (16, 100, 512, 256) if os.environ.get("BILSTM_BENCH_MODEL") == "teacher" else (32, 100, 256, 128)
    st = ops.status_word(dev)
    lens = torch.randint(60, 101, (B,), generator=g).to(torch.int32)
    lens[0] = T
    x = torch.randn(B * T, C, generator=g).to(dev)
    w = [(torch.randn(4 * H, d, generator=g) * 0.05).to(dev) for d in (C, H, C, H)]
    bs = [(torch.randn(4 * H, generator=g) * 0.1).to(dev) for _ in range(2)]
    ld = lens.to(dev)

    def timed(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    out = ops.bilstm(x, ld, w[0], w[1], bs[0], w[2], w[3], bs[1], B, T, 3 if H == 256 else 2)
    t_fwd = timed(lambda: ops.bilstm(x, ld, w[0], w[1], bs[0], w[2], w[3], bs[1], B, T, 3 if H == 256 else 2))
    # the input projections are inside that call: time them alone and subtract
    gx = [torch.randn(B * T, 4 * H, generator=g).to(dev) for _ in range(2)]
    s = [[torch.empty(T * B, 4 * H, device=dev)] + [torch.empty(T * B, H, device=dev) for _ in range(3)] for _ in range(2)]
    o2 = torch.empty(B * T, 2 * H, device=dev)
    t_train = timed(lambda: ops.bilstm_train_fwd(gx, (w[1], w[3]), ld, B, T, o2, s, status=st))
    d_out = torch.randn(B * T, 2 * H, generator=g).to(dev)
    wt = [w[1].t().contiguous(), w[3].t().contiguous()]
    dg = [torch.empty(T * B, 4 * H, device=dev) for _ in range(2)]
    t_bptt = timed(lambda: ops.bilstm_bptt(s, ld, B, T, d_out, wt, dg, status=st))
    torch.cuda.synchronize()
    assert int(st.item()) == 0, int(st.item())
    print("ksplit=%s  fwd (incl. 2 input-projection GEMMs) %.1f us   train fwd %.1f us   bptt %.1f us" % (os.environ.get("FCL_BILSTM_KSPLIT", "1"), t_fwd, t_train, t_bptt))
    return out.cpu(), o2.cpu(), [d.cpu() for d in dg], [[t.cpu() for t in sd] for sd in s]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        r = run()
        torch.save(r, sys.argv[2])
    else:
        import tempfile
        tmp = tempfile.mkdtemp()
        res = {}
        for k in ("0", "1"):
            env = dict(os.environ, FCL_BILSTM_KSPLIT=k, FCL_BILSTM_GROUP_LL=k)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", os.path.join(tmp, k + ".pt")], env=env, check=True)
            res[k] = torch.load(os.path.join(tmp, k + ".pt"))
        a, b = res["0"], res["1"]
        print("max |old - new|: fwd %.2e  train out %.2e  dg %.2e %.2e  saved gates %.2e" % (
            (a[0] - b[0]).abs().max(), (a[1] - b[1]).abs().max(), (a[2][0] - b[2][0]).abs().max(), (a[2][1] - b[2][1]).abs().max(),
            max((x - y).abs().max() for sa, sb in zip(a[3], b[3]) for x, y in zip(sa, sb))))
