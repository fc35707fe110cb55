"""Where a KD update's wall time goes on the STUDENT's main stream, with the frozen teacher one batch ahead on its own stream (developer aid):
FCL_TE_STAMPS=1 python tools/kd_phases.py [kd|teacher]  -> ms per phase (HIP events at the native step's phase boundaries), mean of 20 updates."""
import ctypes as C, os, sys
os.environ.setdefault("FCL_TE_STAMPS", "1")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fcl_taco2_amd  # noqa
from fcl_taco2_amd import hparams as HP, synthetic as SYN
from fcl_taco2_amd.converter import CustomConverter
from fcl_taco2_amd.training import KDPipeline, TrainEngine

what = sys.argv[1] if len(sys.argv) > 1 else "kd"
S, T = HP.student_hparams(), HP.teacher_hparams()
bs = []
for sd in range(4):
    xs, ys, ds, f0, en = SYN.training_batch(80, S.idim, batch=32 if what == "kd" else 16, t_lo=60, t_hi=100, seed=100 + sd, zero_frac=0.03, lam=10.0, hi=50)
    bs.append(CustomConverter(1, True, True)([(xs, ys, None, ds, f0, en)]))
if what == "kd":
    teng = TrainEngine(SYN.build_model("kd_teacher", T, None, "cuda:0"))
    eng = TrainEngine(SYN.build_model("student", S, T, "cuda:0"))
    pipe = KDPipeline(teng, eng)
    step = lambda i: pipe.step(bs[i % 4], bs[(i + 1) % 4])
else:
    eng = TrainEngine(SYN.build_model("teacher", T, None, "cuda:0"))
    step = lambda i: eng.train_step(bs[i % 4], mode="train")
for i in range(5):
    step(i)
torch.cuda.synchronize()
names = ["start", "encoder", "prenet+hoists", "decoder cells", "forward end (postnet)", "losses", "bwd0 postnet", "bwd1 decoder BPTT + prenet", "bwd2 embeds", "bwd3 encoder", "join"]
acc, n = np.zeros(12), 0
import time
t0 = time.perf_counter()
for i in range(20):
    step(i)
    out = (C.c_float * 12)()
    eng.native.lib.fcl_te_phase_ms(eng.native.h, out)
    acc += np.array(list(out)); n += 1
torch.cuda.synchronize()
print("%s update: %.2f ms wall per update (stamped run: one synchronisation per update)" % (what, 1e3 * (time.perf_counter() - t0) / 20))
prev = 0.0
for i, nm in enumerate(names):
    v = acc[i] / n
    if i and v >= 0:
        print("  %-28s +%.3f ms  (at %.3f)" % (nm, v - prev, v))
        prev = v
