"""KD / teacher training step timing on one GPU: python tools/train_bench.py [--role student|teacher] [--batch 32] [--steps 5]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import fcl_taco2_amd  # noqa: E402,F401
from fcl_taco2_amd import _lib, hparams as HP, synthetic as SYN  # noqa: E402
from fcl_taco2_amd.converter import CustomConverter  # noqa: E402
from fcl_taco2_amd.training import TrainEngine  # noqa: E402


def build(role, hp, thp=None, dev="cuda:0"):
    return SYN.build_model(role, hp, thp, dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--role", default="student")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    S, T = HP.student_hparams(), HP.teacher_hparams()
    xs, ys, ds, f0, en = SYN.training_batch(80, S.idim, batch=a.batch, t_lo=60, t_hi=100, seed=1234, zero_frac=0.03, lam=10.0, hi=50)
    batch = CustomConverter(1, True, True)([(xs, ys, None, ds, f0, en)])
    frames = int(sum(y.shape[0] for y in ys))
    if a.role == "student":
        teng = TrainEngine(build("kd_teacher", T))
        eng = TrainEngine(build("student", S, T))
    else:
        teng, eng = None, TrainEngine(build("teacher", T))

    def step():
        know = teng.knowledge(batch, mode="train") if teng else None
        return eng.train_step(batch, know, mode="train")

    for _ in range(a.warmup):
        rep = step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.steps):
        rep = step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / a.steps
    print("role %s  B=%d frames=%d  step %.1f ms  (%.0f frames/s)  loss %.4f gn %.3f" % (a.role, a.batch, frames, dt * 1e3, frames / dt, rep["loss"], rep["grad_norm"]))


if __name__ == "__main__":
    main()
