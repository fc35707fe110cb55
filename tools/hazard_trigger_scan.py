"""Round 6: WHICH foreign work makes the lane-split BiLSTM kernels go wrong?  The victim is the H = 256 cooperating-workgroup forward of a build WITHOUT the exclusive
register file (tools/ab/libfcl_ksne_dpp.so via FCL_LIB, FCL_KS_GUARD=0: wrong in ~100 % of the launches beside `fcl_bilstm_fwd` of another stream); the other stream runs
ONE kind of work at a time.  Prints mismatching iterations per kind."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fcl_taco2_amd  # noqa
from fcl_taco2_amd import ops

dev = "cuda:0"
g = torch.Generator().manual_seed(3)


def case(B, T, C, H):
    lens = torch.randint(60, T + 1, (B,), generator=g).to(torch.int32)
    lens[0] = T
    x = torch.randn(B * T, C, generator=g).to(dev)
    w = [(torch.randn(4 * H, d, generator=g) * 0.05).to(dev) for d in (C, H, C, H)]
    bs = [(torch.randn(4 * H, generator=g) * 0.1).to(dev) for _ in range(2)]
    return dict(B=B, T=T, H=H, x=x, w=w, bs=bs, ld=lens.to(dev))


st = ops.status_word(dev)
fwd = lambda c: ops.bilstm(c["x"], c["ld"], c["w"][0], c["w"][1], c["bs"][0], c["w"][2], c["w"][3], c["bs"][1], c["B"], c["T"], 3 if c["H"] == 256 else 2, status=st)
big, small = case(8, 100, 512, 256), case(8, 100, 256, 128)
ref = fwd(big).clone()
torch.cuda.synchronize()
X = torch.randn(800, 256, device=dev)
W = torch.randn(512, 256, device=dev) * 0.05
Xp, Wp = ops.pack_planes(X), ops.pack_planes(W)
big_buf = torch.zeros(8 << 20, device=dev)
seg_lo = torch.zeros(800, dtype=torch.int32, device=dev)
seg_hi = torch.full((800,), 800, dtype=torch.int32, device=dev)
gam, bet = torch.ones(256, device=dev), torch.zeros(256, device=dev)
kinds = {
    "nothing (victim alone)": lambda: None,
    "fcl_bilstm_fwd H=128 (round-5 stress)": lambda: [fwd(small) for _ in range(3)],
    "pgemm on planes (LDS-DMA GEMM)": lambda: [ops.linear_planes(Xp, Wp, 512, 256) for _ in range(12)],
    "gemm_kernel on fp32 operands (no LDS-DMA)": lambda: [ops.linear(X, W) for _ in range(12)],
    "pack_planes": lambda: [ops.pack_planes(X) for _ in range(24)],
    "layernorm": lambda: [ops.layernorm(X, gam, bet, 1e-5) for _ in range(24)],
    "torch elementwise add": lambda: [big_buf.add_(1.0) for _ in range(24)],
    "memset (tensor.zero_)": lambda: [big_buf.zero_() for _ in range(24)],
    "small memset x many (hipMemsetAsync-like)": lambda: [big_buf[:256].zero_() for _ in range(60)],
    "device-to-device copy": lambda: [big_buf[: 1 << 20].copy_(big_buf[1 << 20 : 2 << 20]) for _ in range(24)],
}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
iters = int(os.environ.get("SCAN_ITERS", "60"))
for name, work in kinds.items():
    bad = 0
    worst = 0.0
    for it in range(iters):
        if it % 2:
            with torch.cuda.stream(s2):
                work()
            with torch.cuda.stream(s1):
                o = fwd(big)
        else:
            with torch.cuda.stream(s1):
                o = fwd(big)
            with torch.cuda.stream(s2):
                work()
        torch.cuda.synchronize()
        e = float((o - ref).abs().max())
        bad += e > 0
        worst = max(worst, e)
    print("%-46s mismatching iterations %3d / %d   worst |diff| %.1e   status %d" % (name, bad, iters, worst, int(st.item())))
