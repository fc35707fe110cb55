"""fp32-equivalent vs bf16 (amp) gradients of one training step on the same batch / masks (developer tool): cosine, norms, worst parameters."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import fcl_taco2_amd  # noqa
from fcl_taco2_amd import hparams as HP, synthetic as SYN
from fcl_taco2_amd.converter import CustomConverter
from fcl_taco2_amd.training import TrainEngine

dev = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "teacher"
S, T = HP.student_hparams(), HP.teacher_hparams()
B = 16 if which == "teacher" else 32
xs, ys, ds, f0, en = SYN.training_batch(80, S.idim, batch=B, t_lo=60, t_hi=100, seed=1234, zero_frac=0.03, lam=10.0, hi=50)
batch = CustomConverter(1, True, True)([(xs, ys, None, ds, f0, en)])
res = {}
for amp in (None, "bf16"):
    if which == "teacher":
        eng = TrainEngine(SYN.build_model("teacher", T, None, dev), seed=0, amp=amp)
        know = None
    else:
        teng = TrainEngine(SYN.build_model("kd_teacher", T, None, dev), amp=amp)
        know = teng.knowledge(batch, mode="train")
        eng = TrainEngine(SYN.build_model("student", S, T, dev), seed=0, amp=amp)
    eng.zero_grad()
    rep = eng.forward_backward(batch, know, mode="train")
    torch.cuda.synchronize()
    res[amp] = (eng.gflat.clone(), {k: float(v) for k, v in rep.items() if isinstance(v, (int, float))}, eng)
g0, g1 = res[None][0].double(), res["bf16"][0].double()
print("loss", res[None][1].get("loss"), res["bf16"][1].get("loss"))
print("norms", float(g0.norm()), float(g1.norm()), "cosine", float((g0 * g1).sum() / g0.norm() / g1.norm()))
eng = res[None][2]
rows = []
for k, (o, n, shp) in eng.param_offsets().items():
    a, b = g0[o:o + n], g1[o:o + n]
    rows.append((float((a - b).norm()), float(a.norm()), float(b.norm()), k))
rows.sort(reverse=True)
for r in rows[:12]:
    print("|d| %.4f  |g32| %.4f  |gbf| %.4f  %s" % r)
