"""Developer aid: is a training update bound by the submitting thread?  Times the host's enqueue of K updates (no synchronisation inside) against the
wall time of the same K updates: host ~ wall -> the GPU waits for launches; host << wall -> the GPU's own time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcl_taco2_amd import hparams as HP, synthetic as SYN
from fcl_taco2_amd.converter import CustomConverter
from fcl_taco2_amd.training import TrainEngine, KDPipeline

kd = (sys.argv[1] if len(sys.argv) > 1 else "kd") == "kd"
dev = torch.device("cuda:0")
S, T = HP.student_hparams(), HP.teacher_hparams()
B = 32 if kd else 16
torch.set_num_threads(4)
xs, ys, ds, f0, en = SYN.training_batch(80, S.idim, batch=B, t_lo=60, t_hi=100, seed=1234, zero_frac=0.03, lam=10.0, hi=50)
batch = CustomConverter(1, True, True)([(xs, ys, None, ds, f0, en)])
for k in ("xs", "ys", "extras", "f0", "energy"):
    batch[k] = batch[k].to(dev)
teng = TrainEngine(SYN.build_model("kd_teacher", T, None, dev)) if kd else None
eng = TrainEngine(SYN.build_model("student", S, T, dev) if kd else SYN.build_model("teacher", T, None, dev))
pipe = KDPipeline(teng, eng) if kd else None
batches = [batch, dict(batch)]
def step(i):
    if pipe is not None:
        return pipe.step(batches[i % 2], batches[(i + 1) % 2])
    return eng.train_step(batch, None, mode="train")
for i in range(5):
    step(i)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for i in range(K):
    step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
# the enqueue of ONE update on an idle queue (round 4: with the native step the host runs far ahead of the device, so K back-to-back enqueues
# measure the HIP queue's back-pressure, i.e. the device, not the host)
one = []
for i in range(9):
    torch.cuda.synchronize()
    a = time.perf_counter()
    step(i)
    one.append(time.perf_counter() - a)
torch.cuda.synchronize()
one.sort()
print("host enqueue of one update (idle queue, median of 9) %.2f ms; %d back-to-back updates: host %.2f ms / update, wall %.2f ms / update; native step: %s"
      % (one[4] * 1e3, K, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3, eng.native is not None))
