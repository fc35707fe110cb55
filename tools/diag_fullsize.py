"""Diagnostic: per-tensor gradient error of a full-size teacher training step against the oracle in FLOAT64 (the truth), next to the fp32 oracle's
own error against it (the noise floor of an fp32 implementation).  usage: diag_fullsize.py [B] [eval|train] [closed_form|init]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import fcl_oracle as O  # noqa: E402
import test_gpu_training_fullsize as TF  # noqa: E402
from fcl_taco2_amd import hparams as HP, synthetic as SYN  # noqa: E402
from fcl_taco2_amd.training import TrainEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
form = sys.argv[2] if len(sys.argv) > 2 else "eval"
weights = sys.argv[3] if len(sys.argv) > 3 else "closed_form"
TF._threads()
mk = HP.student_hparams if os.environ.get("DIMS", "T") == "S" else HP.teacher_hparams
T = mk() if form == "train" else mk(dropout_rate=0.0)
TOP = int(os.environ.get("TOP", "30"))
batch = TF._batch(B, 41, T.idim)
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


r32, s32 = oracle(torch.float32)
r64, s64 = oracle(torch.float64)
for k in TF.LOSS_KEYS:
    print("%-12s hip %.7f  oracle32 %.7f  oracle64 %.7f" % (k, rep[k], float(r32[k]), float(r64[k])))
rows = []
for k, v in s64.items():
    if v.dtype.is_floating_point and v.requires_grad:
        ref = v.grad if v.grad is not None else torch.zeros_like(v)
        den_l2 = max(float(ref.norm()), 1e-3 * ref.numel() ** 0.5)
        den_mx = max(1.0, float(ref.abs().max()))
        g = eng.G[k].cpu().double()
        o = s32[k].grad.double() if s32[k].grad is not None else torch.zeros_like(ref)
        rows.append((float((g - ref).norm()) / den_l2, float((g - ref).abs().max()) / den_mx, float((o - ref).norm()) / den_l2,
                     float((o - ref).abs().max()) / den_mx, k, float(ref.abs().max())))
print("%-9s %-9s | %-9s %-9s  (HIP L2, max | fp32-oracle L2, max; relative to the fp64 oracle)" % ("hipL2", "hipMax", "o32L2", "o32Max"))
for r in sorted(rows, reverse=True)[:TOP]:
    print("%.3e %.3e | %.3e %.3e  %-45s max|ref| %.3e" % r)
