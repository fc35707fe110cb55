"""Central finite difference of the KD loss along the gradient at several step sizes, repeated (developer aid for the bound in
tests/test_gpu_training.py::test_full_size_kd_step_properties: the loss carries ~2e-5 of summation-order noise)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fcl_taco2_amd  # noqa
from fcl_taco2_amd import hparams as HP, synthetic as SYN
from fcl_taco2_amd.converter import CustomConverter
from fcl_taco2_amd.training import TrainEngine

DEV = "cuda:0"
S, T = HP.student_hparams(dropout_rate=0.0), HP.teacher_hparams(dropout_rate=0.0)
xs, ys, ds, f0, en = SYN.training_batch(80, S.idim, batch=8, t_lo=60, t_hi=100, seed=77, zero_frac=0.03, lam=10.0, hi=50)
batch = CustomConverter(1, True, True)([(xs, ys, None, ds, f0, en)])
teacher = SYN.build_model("kd_teacher", T, None, DEV).eval()
student = SYN.build_model("student", S, T, DEV).eval()
with torch.no_grad():
    know = teacher(**{k: v for k, v in batch.items()})
eng = TrainEngine(student)
eng.forward_backward(batch, teacher_knowledge=know)
for name in ("dec.feat_out.weight", "enc.convs.1.0.weight", "dec.lstm_proj.weight"):
    g = eng.G[name].clone()
    gnorm = float(g.norm())
    d = g / gnorm
    w0 = eng.P[name].clone()
    for eps in (2e-3, 4e-3, 8e-3, 1.6e-2, 3.2e-2):
        fds = []
        for rep in range(5):
            vals = []
            for sgn in (+1.0, -1.0):
                eng.P[name].copy_(w0 + sgn * eps * d)
                eng.zero_grad()
                vals.append(eng.forward_backward(batch, teacher_knowledge=know)["loss"])
            fds.append((vals[0] - vals[1]) / (2 * eps))
        eng.P[name].copy_(w0)
        fds = np.array(fds)
        print("%-24s |g| %.5f  eps %.1e: fd mean %.5f  spread %.1e  rel err of the mean %.2e  worst single %.2e" % (
            name, gnorm, eps, fds.mean(), fds.max() - fds.min(), abs(fds.mean() - gnorm) / gnorm, np.abs(fds - gnorm).max() / gnorm))
