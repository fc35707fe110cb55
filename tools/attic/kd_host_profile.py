"""Developer aid: cProfile of the submitting thread over a few KD updates (where the 16 us per launch go)."""
import cProfile, pstats, os, sys, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcl_taco2_amd import hparams as HP, synthetic as SYN
from fcl_taco2_amd.converter import CustomConverter
from fcl_taco2_amd.training import TrainEngine, KDPipeline
dev = torch.device("cuda:0")
S, T = HP.student_hparams(), HP.teacher_hparams()
torch.set_num_threads(4)
xs, ys, ds, f0, en = SYN.training_batch(80, S.idim, batch=32, t_lo=60, t_hi=100, seed=1234, zero_frac=0.03, lam=10.0, hi=50)
batch = CustomConverter(1, True, True)([(xs, ys, None, ds, f0, en)])
for k in ("xs", "ys", "extras", "f0", "energy"):
    batch[k] = batch[k].to(dev)
teng = TrainEngine(SYN.build_model("kd_teacher", T, None, dev))
eng = TrainEngine(SYN.build_model("student", S, T, dev))
pipe = KDPipeline(teng, eng)
batches = [batch, dict(batch)]
for i in range(5):
    pipe.step(batches[i % 2], batches[(i + 1) % 2])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    pipe.step(batches[i % 2], batches[(i + 1) % 2])
pr.disable()
torch.cuda.synchronize()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(28)
print(out.getvalue()[:6000])
