"""Diagnostic: one predictor's forward + backward on the HIP engine against the float64 oracle, isolated from the rest of the model."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import fcl_oracle as O  # noqa: E402
import test_gpu_training_fullsize as TF  # noqa: E402
from fcl_taco2_amd import hparams as HP, synthetic as SYN  # noqa: E402
from fcl_taco2_amd.training import TrainEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = HP.teacher_hparams()
batch = TF._batch(B, 41, T.idim)
masks = TF.random_masks(T, batch, 7)
model = SYN.build_model("teacher", T, None, "cuda:0", weights="init", seed=1)
eng = TrainEngine(model, overlap_dw=False)
c = eng._ctx(batch, "train", masks)
ilens = [int(v) for v in batch["ilens"]]
Bn, Tm = len(ilens), max(ilens)
rng = np.random.RandomState(5)
hs_np = rng.randn(Bn, Tm, T.eunits).astype(np.float32) * 0.5
for b, n in enumerate(ilens):
    hs_np[b, n:] = 0.0  # pad_packed_sequence zeros
dout_np = rng.randn(Bn, Tm).astype(np.float32)
hs_dev = torch.from_numpy(hs_np.reshape(Bn * Tm, -1)).cuda()
sd64 = {k: (v.detach().cpu().double().clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.detach().cpu())
        for k, v in model.state_dict().items()}
pad = O.make_pad_mask(ilens)
for name, layers, p in (("duration_predictor", T.duration_predictor_layers, T.duration_predictor_dropout_rate),
                        ("pitch_predictor", T.variance_predictor_layers, T.variance_predictor_dropout_rate),
                        ("energy_predictor", T.variance_predictor_layers, T.variance_predictor_dropout_rate)):
    eng.zero_grad()
    out, caches = eng._predictor_fwd(c, hs_dev, name, layers, p, c.e_lo, c.e_hi, c.enc_pad)
    dx = eng._predictor_bwd(torch.from_numpy(dout_np.reshape(-1)).cuda(), name, caches, c.enc_pad)
    eng._join_dw()
    torch.cuda.synchronize()
    hs64 = torch.from_numpy(hs_np).double().requires_grad_(True)
    if name == "duration_predictor":
        o = O.duration_predictor(sd64, T, hs64, pad, keeps=masks[name])
    else:
        o = O.variance_predictor(sd64, T, name.split("_")[0], hs64, pad, masks[name]).squeeze(-1)
    for v in sd64.values():
        if torch.is_tensor(v) and v.grad is not None:
            v.grad = None
    (o * torch.from_numpy(dout_np).double()).sum().backward()
    rel = lambda a, r: float((a.detach().cpu().double().reshape(r.shape) - r).norm() / r.norm())
    print("%-20s B=%d  out %.2e  d_hs %.2e  " % (name, B, rel(out.reshape(Bn, Tm), o.detach()), rel(dx, hs64.grad)) +
          "  ".join("%s %.1e" % (k[len(name) + 1:], rel(eng.G[k], sd64[k].grad)) for k in sorted(sd64) if k.startswith(name + ".")))
    # ---- the last LayerNorm's backward alone, on the HIP path's own input y (isolates the kernel from forward differences)
    cc, y, i, last, keep, ks = caches[-1]
    from fcl_taco2_amd import ops
    from fcl_taco2_amd.plan import LN_EPS
    P = eng.P
    g_, b_ = P["%s.conv.%d.2.weight" % (name, i)], P["%s.conv.%d.2.bias" % (name, i)]
    dg, db, dlw, dlb = [torch.zeros_like(t) for t in (g_, b_, g_, P[name + ".linear.bias"])]
    dyh = ops.layernorm_bwd(y, g_, b_, LN_EPS, dg, db, lin_w=P[name + ".linear.weight"].reshape(-1), ds=torch.from_numpy(dout_np.reshape(-1)).cuda(),
                            pad_mask=c.enc_pad, dlin_w=dlw, dlin_b=dlb, keep=keep, keep_scale=ks)
    for dt in (torch.float64, torch.float32):
        y64 = y.detach().cpu().to(dt).requires_grad_(True)
        ln = torch.nn.functional.layer_norm(y64, (y64.shape[1],), g_.detach().cpu().to(dt), b_.detach().cpu().to(dt), 1e-12)
        if keep is not None:
            ln = ln * keep.cpu().to(dt) * ks
        s = ln @ P[name + ".linear.weight"].detach().cpu().to(dt).reshape(-1) + P[name + ".linear.bias"].detach().cpu().to(dt)
        s = s.masked_fill(c.enc_pad.cpu().bool(), 0.0)
        (s * torch.from_numpy(dout_np.reshape(-1)).to(dt)).sum().backward()
        if dt == torch.float64:
            ref = y64.grad.clone()
            var = y.detach().cpu().double().var(dim=1, unbiased=False)
            err_row = (dyh.cpu().double() - ref).norm(dim=1)
            worst = int(err_row.argmax())
            print("   LN bwd alone: HIP vs fp64 %.2e ; worst row %d: err %.2e |ref row| %.2e var %.3e min var %.3e" %
                  (rel(dyh, ref), worst, float(err_row[worst]), float(ref[worst].norm()), float(var[worst]), float(var.min())))
        else:
            print("   LN bwd alone: torch fp32 vs fp64 %.2e" % rel(y64.grad, ref))
    # ---- ReLU mask flips between the HIP forward and the float64 forward (a flipped element changes the gradient by O(1), not by rounding)
    F_ = torch.nn.functional
    x = torch.from_numpy(hs_np).double()
    for li in range(layers):
        w, bb = sd64["%s.conv.%d.0.weight" % (name, li)].detach(), sd64["%s.conv.%d.0.bias" % (name, li)].detach()
        pre = F_.conv1d(x.transpose(1, 2), w, bb, 1, (w.shape[-1] - 1) // 2).transpose(1, 2)
        y_h = caches[li][1].detach().cpu().double().reshape(pre.shape)
        flips = ((y_h > 0) != (pre > 0))
        print("   layer %d: %d ReLU mask flips of %d; |pre| at flips %s ; max |y_hip - relu(pre)| %.2e" %
              (li, int(flips.sum()), flips.numel(), [float("%.2e" % v) for v in pre[flips].abs()[:5]], float((y_h - pre.clamp(min=0)).abs().max())))
        ln = F_.layer_norm(pre.clamp(min=0), (pre.shape[-1],), sd64["%s.conv.%d.2.weight" % (name, li)].detach(), sd64["%s.conv.%d.2.bias" % (name, li)].detach(), 1e-12)
        x = ln * torch.from_numpy(np.asarray(masks[name][li])).double() / (1.0 - p)
