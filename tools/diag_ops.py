"""Diagnostic: RELATIVE L2 error of the backward building blocks against float64 torch at training sizes (rows = 16 utterances x 100)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import fcl_taco2_amd  # noqa: E402
from fcl_taco2_amd import ops  # noqa: E402

DEV = "cuda:0"
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
rel = lambda a, ref: float((a.detach().cpu().double() - ref.double()).norm() / ref.double().norm())
B, TT = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 100
M = B * TT
rng = np.random.RandomState(0)
lo = np.repeat(np.arange(B) * TT, TT).astype(np.int32)
hi = (lo + TT).astype(np.int32)
for cin, cout, k in ((512, 512, 5), (512, 384, 3), (384, 384, 3)):
    x = torch.from_numpy(rng.randn(M, cin)).requires_grad_(True)
    w = torch.from_numpy(rng.randn(cout, cin, k) / np.sqrt(cin * k)).requires_grad_(True)
    dy = torch.from_numpy(rng.randn(M, cout))
    y = F.conv1d(x.reshape(B, TT, cin).transpose(1, 2), w, None, 1, (k - 1) // 2).transpose(1, 2).reshape(M, cout)
    (y * dy).sum().backward()
    wp = ops.pack_conv1d_weight(dev(w.detach().float().numpy()))
    yh = ops.conv1d(dev(x.detach().float().numpy()), wp, None, dev(lo), dev(hi))
    wt = torch.stack([ops.transpose2d(wp[k - 1 - j]) for j in range(k)]).contiguous()
    dx = ops.conv1d(dev(dy.float().numpy()), wt, None, dev(lo), dev(hi))
    dwp = torch.zeros(k, cout, cin, device=DEV)
    ops.gemm_tn_taps(dev(dy.float().numpy()), dev(x.detach().float().numpy()), dwp, -(k - 1) // 2, dev(lo), dev(hi))
    print("conv %dx%dx%d M=%d: y %.2e  dx %.2e  dW %.2e" % (cin, cout, k, M, rel(yh, y), rel(dx, x.grad), rel(dwp.permute(1, 2, 0), w.grad)))
for c in (384,):
    x0 = np.maximum(rng.randn(M, c), 0) * 1.3
    g0, b0, lw0, lb0 = 1 + 0.1 * rng.randn(c), 0.1 * rng.randn(c), rng.randn(c) / np.sqrt(c), rng.randn(1)
    dy, ds = rng.randn(M, c), rng.randn(M)
    keep = (rng.rand(M, c) < 0.5).astype(np.uint8)
    x, g, b, lw, lb = [torch.from_numpy(v).requires_grad_(True) for v in (x0, g0, b0, lw0, lb0)]
    yy = F.layer_norm(x, (c,), g, b, 1e-12) * torch.from_numpy(keep).double() * 2.0
    s = yy @ lw + lb
    ((yy * torch.from_numpy(dy)).sum() + (s * torch.from_numpy(ds)).sum()).backward()
    f = lambda a: dev(np.asarray(a, np.float32))
    dg, db, dlw, dlb = (torch.zeros(c, device=DEV), torch.zeros(c, device=DEV), torch.zeros(c, device=DEV), torch.zeros(1, device=DEV))
    dx = ops.layernorm_bwd(f(x0), f(g0), f(b0), 1e-12, dg, db, dy=f(dy), lin_w=f(lw0), ds=f(ds), dlin_w=dlw, dlin_b=dlb, keep=dev(keep), keep_scale=2.0)
    print("layernorm_bwd C=%d M=%d: dx %.2e dgamma %.2e dbeta %.2e dlw %.2e" % (c, M, rel(dx, x.grad), rel(dg, g.grad), rel(db, b.grad), rel(dlw, lw.grad)))
for c in (512,):
    z0 = rng.randn(M, c) * 0.7 + 0.3
    g0, b0 = 1 + 0.1 * rng.randn(c), 0.1 * rng.randn(c)
    dy = rng.randn(M, c) + 0.5
    z, g, b = [torch.from_numpy(v).requires_grad_(True) for v in (z0, g0, b0)]
    yy = F.batch_norm(z, None, None, g, b, True, 0.1, 1e-5)
    (yy * torch.from_numpy(dy)).sum().backward()
    f = lambda a: dev(np.asarray(a, np.float32))
    zz = f(z0)
    mean, invstd = ops.bn_stats(zz, 1e-5)
    dbeta, dgamma = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    ops.colsum(f(dy), dbeta)
    ops.colsum(f(dy), dgamma, y=zz, gamma=invstd, beta=mean, mode=3)
    dz = ops.bn_bwd(f(dy), zz, mean, invstd, f(g0), dbeta, dgamma)
    print("batchnorm train bwd C=%d M=%d: dz %.2e dgamma %.2e dbeta %.2e" % (c, M, rel(dz, z.grad), rel(dgamma, g.grad), rel(dbeta, b.grad)))
