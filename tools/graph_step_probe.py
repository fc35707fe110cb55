"""Probe (round 5): what a training update costs as ONE hipGraph launch -- the main stream of the native step is 231 dependent launches with a median gap of
~7 us between them (tools/trace_kd.sh teacher_step); inside a captured graph the gap is ~1.5 us.  Captures one update (same batch, device RNG draws baked at
capture time) and replays it: the time a per-step pipelined capture (capture batch i + 1 on the host while the device runs batch i) could reach.
Usage: python tools/graph_step_probe.py [teacher|kd]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fcl_taco2_amd  # noqa
from fcl_taco2_amd import hparams as HP, synthetic as SYN
from fcl_taco2_amd.converter import CustomConverter
from fcl_taco2_amd.training import TrainEngine

what = sys.argv[1] if len(sys.argv) > 1 else "teacher"
S, T = HP.student_hparams(), HP.teacher_hparams()
xs, ys, ds, f0, en = SYN.training_batch(80, S.idim, batch=32 if what == "kd" else 16, t_lo=60, t_hi=100, seed=100, zero_frac=0.03, lam=10.0, hi=50)
batch = CustomConverter(1, True, True)([(xs, ys, None, ds, f0, en)])
if what == "kd":
    teng = TrainEngine(SYN.build_model("kd_teacher", T, None, "cuda:0"))
    eng = TrainEngine(SYN.build_model("student", S, T, "cuda:0"))
    know = teng.knowledge(batch, mode="train", native=True)
else:
    eng = TrainEngine(SYN.build_model("teacher", T, None, "cuda:0"))
    know = None


def update():
    eng.zero_grad()
    eng.forward_backward(batch, know, "train", None)
    eng.optimizer_step()


def timed(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(5):
    update()
print("%s update, launch by launch: %.3f ms" % (what, timed(update)))
g = torch.cuda.CUDAGraph()
t0 = time.perf_counter()
with torch.cuda.graph(g, capture_error_mode="relaxed"):
    update()
t1 = time.perf_counter()
print("stream capture + instantiate on the host: %.2f ms" % ((t1 - t0) * 1e3))
for _ in range(3):
    g.replay()
print("%s update as one graph launch: %.3f ms" % (what, timed(g.replay)))
