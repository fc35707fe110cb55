"""Developer aid: is the KD update host-bound?  Enqueue time of N updates (before the final synchronize) vs their total time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fcl_taco2_amd  # noqa
from fcl_taco2_amd import hparams as HP, synthetic as SYN
from fcl_taco2_amd.converter import CustomConverter
from fcl_taco2_amd.training import TrainEngine, KDPipeline

torch.set_num_threads(4)
dev = torch.device("cuda:0")
S, T = HP.student_hparams(), HP.teacher_hparams()
xs, ys, ds, f0, en = SYN.training_batch(80, S.idim, batch=32, t_lo=60, t_hi=100, seed=1234, zero_frac=0.03, lam=10.0, hi=50)
batch = CustomConverter(1, True, True)([(xs, ys, None, ds, f0, en)])
for k in ("xs", "ys", "extras", "f0", "energy"):
    batch[k] = batch[k].to(dev)
teng = TrainEngine(SYN.build_model("kd_teacher", T, None, dev))
eng = TrainEngine(SYN.build_model("student", S, T, dev))
pipe = KDPipeline(teng, eng)
batches = [batch, dict(batch)]
for i in range(3):
    pipe.step(batches[i % 2], batches[(i + 1) % 2])
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
for i in range(N):
    pipe.step(batches[i % 2], batches[(i + 1) % 2])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.2f ms/update, total %.2f ms/update" % (1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N))
# student alone, teacher alone
know = teng.knowledge(batch, mode="train")
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(N):
    eng.train_step(batch, know, mode="train")
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("student alone: enqueue %.2f, total %.2f ms" % (1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N))
t0 = time.perf_counter()
for i in range(N):
    teng.knowledge(batch, mode="train")
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("teacher alone: enqueue %.2f, total %.2f ms" % (1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N))

import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for i in range(5):
    eng.train_step(batch, know, mode="train")
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
