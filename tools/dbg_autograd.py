import os, sys
import torch
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import test_gpu_training as TG
from helpers import TINY_T7
from fcl_taco2_amd.training import TrainEngine
batch = TG._batch()
kw = {k: v for k, v in batch.items() if not k.startswith("_")}
lr = 2e-3
ma = TG._model("teacher", TINY_T7).train()
ea = ma.train_engine(seed=3)
opt = torch.optim.SGD(ma.parameters(), lr=lr)
eb = TrainEngine(TG._model("teacher", TINY_T7), seed=3)
for it in range(3):
    opt.zero_grad()
    loss = ma(**kw)
    loss.backward()
    ga = {k: p.grad.clone() for k, p in ma.named_parameters()}
    pa = {k: p.detach().clone() for k, p in ma.named_parameters()}
    eb.zero_grad()
    lb = eb.forward_backward(batch, mode="train", reduce=False)["loss"]
    gb = {k: eb.G[k].clone() for k in eb.G}
    pb = {k: eb.P[k].clone() for k in eb.P}
    print("step", it, "loss", float(loss), lb, "param max diff", max(float((pa[k] - pb[k]).abs().max()) for k in pa))
    rows = []
    for k in ga:
        a, b = ga[k].double().reshape(-1), gb[k].double().reshape(-1)
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-300))
        rows.append((float((a - b).abs().max()), k, float(b.abs().max()), cos))
    for r in sorted(rows, reverse=True)[:8]:
        print("   %.3e %-45s max|g| %.3e cos %.9f" % r)
    opt.step()
    eb.pflat.add_(eb.gflat, alpha=-lr)
