"""developer aid: per-parameter gradient error of the no-batch-norm teacher step (eval form) against the oracle's autograd"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_training as T
from helpers import TINY_TN, max_abs
from oracle import fcl_oracle as O
from fcl_taco2_amd.training import TrainEngine
batch = T._batch()
eng = TrainEngine(T._model("teacher", TINY_TN))
rep = eng.forward_backward(batch)
sd = T._grad_sd(TINY_TN)
orep = O.model_forward(sd, TINY_TN, T._cpu(batch), "teacher")
orep["loss"].backward()
print("loss", rep["loss"], float(orep["loss"]))
for k, v in sd.items():
    if v.dtype.is_floating_point and v.requires_grad:
        ref = torch.zeros_like(v) if v.grad is None else v.grad
        print("%-45s %.3e  (max |ref| %.3e)" % (k, max_abs(eng.G[k].cpu(), ref), float(ref.abs().max())))
