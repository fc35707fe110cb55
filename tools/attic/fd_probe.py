import os
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_training as T
from fcl_taco2_amd import hparams as HP, synthetic as SYN, teacher_forced as TF
from fcl_taco2_amd.converter import CustomConverter
from fcl_taco2_amd.training import TrainEngine
DEV = "cuda:0"
S, Tt = HP.student_hparams(dropout_rate=0.0), HP.teacher_hparams(dropout_rate=0.0)
xs, ys, ds, f0, en = SYN.training_batch(80, S.idim, batch=8, t_lo=60, t_hi=100, seed=77, zero_frac=0.03, lam=10.0, hi=50)
batch = CustomConverter(1, True, True)([(xs, ys, None, ds, f0, en)])
teacher = SYN.build_model("kd_teacher", Tt, None, DEV).eval()
student = SYN.build_model("student", S, Tt, DEV).eval()
with torch.no_grad():
    know = teacher(**{k: v for k, v in batch.items()})
eng = TrainEngine(student)
for rep_i in range(4):
    eng.zero_grad()
    rep = eng.forward_backward(batch, teacher_knowledge=know)
    ref, _ = TF.student_forward(student.plan(), batch, know, True, dropout_mode=0)
    print("loss keys worst rel", max(abs(rep[k] - ref[k]) / max(1.0, abs(ref[k])) for k in T.KD_KEYS))
    for name in ("dec.feat_out.weight", "enc.convs.1.0.weight", "dec.lstm_proj.weight"):
        g = eng.G[name].clone(); gnorm = float(g.norm()); d = g / gnorm; eps = 2e-3
        w0 = eng.P[name].clone(); vals = []
        for sgn in (+1.0, -1.0):
            eng.P[name].copy_(w0 + sgn * eps * d); eng.zero_grad()
            vals.append(eng.forward_backward(batch, teacher_knowledge=know)["loss"])
        eng.P[name].copy_(w0)
        fd = (vals[0] - vals[1]) / (2 * eps)
        print("  %-24s fd %.5f gnorm %.5f rel err %.4f (tol %.4f)" % (name, fd, gnorm, abs(fd - gnorm) / gnorm, 2e-2 + 1e-3 / gnorm))
    eng.zero_grad(); eng.forward_backward(batch, teacher_knowledge=know)
