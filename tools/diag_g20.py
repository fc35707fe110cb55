"""Per-tensor gradient error of the elayers-2 teacher step (G20) vs the oracle's autograd, in float64 and fp32 (developer aid)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import fcl_oracle as O
from helpers import TINY_VARIANTS
import test_gpu_training as TG
from fcl_taco2_amd.training import TrainEngine

name = sys.argv[1] if len(sys.argv) > 1 else "g20_teacher_elayers2"
hp = TINY_VARIANTS[name]
eng = TrainEngine(TG._model("teacher", hp))
batch = TG._batch()
rep = eng.forward_backward(batch)
g = TG._golden(name)
sd = TG._grad_sd(hp)
sd64 = {k: (v.detach().double().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else (v.double() if v.dtype.is_floating_point else v)) for k, v in sd.items()}
b64 = {k: (v.cpu().double() if torch.is_tensor(v) and v.dtype.is_floating_point else (v.cpu() if torch.is_tensor(v) else v)) for k, v in batch.items()}
o = O.model_forward(sd64, hp, b64, "teacher"); o["loss"].backward()
o32 = O.model_forward(sd, hp, TG._cpu(batch), "teacher"); o32["loss"].backward()
print("loss hip %.6f oracle64 %.6f golden %.6f" % (rep["loss"], float(o["loss"]), float(g["loss"])))
for k in sorted(eng.G):
    ref = sd64[k].grad
    if ref is None: continue
    sc = max(1.0, float(ref.abs().max()))
    e_hip = float((eng.G[k].cpu().double() - ref).abs().max()) / sc
    e_o32 = float((sd[k].grad.double() - ref).abs().max()) / sc if sd[k].grad is not None else -1
    e_gold = float((torch.from_numpy(g["grad:" + k]).double() - ref).abs().max()) / sc if "grad:" + k in g else -1
    if e_hip > 5e-5 or "blstm" in k:
        print("%-45s hip %.2e  oracle32 %.2e  golden %.2e  |ref|max %.3e" % (k, e_hip, e_o32, e_gold, float(ref.abs().max())))
