"""Run-to-run determinism of the KD update's forward (developer aid): the frozen teacher's knowledge alone, the sequential pair, the pipeline.
Fresh engines with the same seeds every time; the first loss of a run must repeat to 1e-9."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fcl_taco2_amd  # noqa
from fcl_taco2_amd import hparams as HP, synthetic as SYN
from fcl_taco2_amd.converter import CustomConverter
from fcl_taco2_amd.training import KDPipeline, TrainEngine

DEV = "cuda:0"
S, T = HP.student_hparams(), HP.teacher_hparams()
bs = []
for sd_ in (5, 6):
    xs, ys, ds, f0, en = SYN.training_batch(80, S.idim, batch=8, t_lo=60, t_hi=100, seed=sd_, zero_frac=0.03, lam=10.0, hi=50)
    bs.append(CustomConverter(1, True, True)([(xs, ys, None, ds, f0, en)]))


def engines():
    return TrainEngine(SYN.build_model("kd_teacher", T, None, DEV), seed=11), TrainEngine(SYN.build_model("student", S, T, DEV), seed=5)


print("env", {k: v for k, v in os.environ.items() if k.startswith("FCL_")})
for rep in range(3):
    teng, eng = engines()
    k = teng.knowledge(bs[0], mode="train", native=False)
    torch.cuda.synchronize()
    print("teacher knowledge rep %d: after %.9e enc4 %.9e dec2 %.9e" % (rep, float(k[0].double().abs().sum()), float(k[2][4].double().abs().sum()), float(k[3][2].double().abs().sum())))
for rep in range(3):
    teng, eng = engines()
    ls = []
    for i in range(3):
        k = teng.knowledge(bs[i % 2], mode="train", native=True)
        torch.cuda.synchronize()
        ls.append(float(eng.train_step(bs[i % 2], k, mode="train")["loss"]))
        torch.cuda.synchronize()
    print("sequential rep %d:" % rep, ["%.12f" % l for l in ls])
for rep in range(4):
    teng, eng = engines()
    pipe = KDPipeline(teng, eng)
    ls = []
    for i in range(3):
        ls.append(float(pipe.step(bs[i % 2], bs[(i + 1) % 2] if i + 1 < 3 else None)["loss"]))
    torch.cuda.synchronize()
    print("pipeline rep %d:" % rep, ["%.12f" % l for l in ls])
