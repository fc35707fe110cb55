"""Diagnostic: per-tensor RELATIVE gradient error of a training step at tiny dims against the fp64 oracle.  usage: diag_tiny.py [eval|train] [role]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import fcl_oracle as O  # noqa: E402
import test_gpu_training_fullsize as TF  # noqa: E402
from helpers import TINY_T, TINY_T7  # noqa: E402
from fcl_taco2_amd import synthetic as SYN  # noqa: E402
from fcl_taco2_amd.converter import CustomConverter  # noqa: E402
from fcl_taco2_amd.training import TrainEngine  # noqa: E402

form = sys.argv[1] if len(sys.argv) > 1 else "eval"
weights = sys.argv[2] if len(sys.argv) > 2 else "init"
T = TINY_T7 if form == "train" else TINY_T
xs, ys, ds, f0, en = SYN.training_batch(T.odim, T.idim, batch=6, t_lo=5, t_hi=12, seed=3, zero_frac=0.1, lam=3.0, hi=8)
batch = CustomConverter(1, True, True)([(xs, ys, None, ds, f0, en)])
model = SYN.build_model("teacher", T, None, "cuda:0", weights=weights, seed=1)
masks = TF.random_masks(T, batch, 7) if form == "train" else None
eng = TrainEngine(model)
rep = eng.forward_backward(batch, mode=form, masks=masks)


def oracle(dt):
    sd = {k: (v.detach().cpu().to(dt).clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else
              (v.detach().cpu().to(dt) if v.dtype.is_floating_point else v.detach().cpu().clone())) for k, v in model.state_dict().items()}
    b = {k: (v.cpu().to(dt) if torch.is_tensor(v) and v.dtype.is_floating_point else (v.cpu() if torch.is_tensor(v) else v)) for k, v in batch.items()}
    r = O.model_forward(sd, T, b, "teacher", bn_train=form == "train", masks=masks)
    r["loss"].backward()
    return r, sd


r64, s64 = oracle(torch.float64)
for k in TF.LOSS_KEYS:
    print("%-12s hip %.8f  oracle64 %.8f" % (k, rep[k], float(r64[k])))
rows = []
for k, v in s64.items():
    if v.dtype.is_floating_point and v.requires_grad:
        ref = v.grad if v.grad is not None else torch.zeros_like(v)
        g = eng.G[k].cpu().double()
        rows.append((float((g - ref).norm()) / max(float(ref.norm()), 1e-30), k, float(ref.norm())))
for r in sorted(rows, reverse=True):
    print("%.3e  %-45s |ref| %.3e" % r)
