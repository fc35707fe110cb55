import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import test_gpu_training as T
from helpers import TINY_TK
import fcl_oracle as O
from fcl_taco2_amd.training import TrainEngine
g = T._golden("g13_teacher_spk")
batch = T._batch()
batch["spembs"] = torch.from_numpy(g["spembs"])
eng = TrainEngine(T._model("teacher", TINY_TK))
rep = eng.forward_backward(batch)
sd = T._grad_sd(TINY_TK)
orep = O.model_forward(sd, TINY_TK, T._cpu(batch), "teacher")
orep["loss"].backward()
print("loss", rep["loss"], float(orep["loss"]), float(g["loss"]))
for k, v in sd.items():
    if v.dtype.is_floating_point and v.requires_grad:
        ref = v.grad if v.grad is not None else torch.zeros_like(v)
        e = float((eng.G[k].cpu() - ref).abs().max())
        gk = "grad:" + k
        eg = float(np.abs(eng.G[k].cpu().numpy() - g[gk]).max()) if gk in g else float("nan")
        if e > 1e-4 or (eg == eg and eg > 1e-4):
            print("%-45s vs oracle %.2e  vs golden %.2e  (max |ref| %.2e)" % (k, e, eg, float(ref.abs().max())))
