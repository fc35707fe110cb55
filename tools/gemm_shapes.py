"""Per-shape HIP-event table of the GEMM launches of one training update (developer aid): FCL_PROF_SHAPES=1 python tools/gemm_shapes.py [kd|teacher]"""
import os, sys
os.environ.setdefault("FCL_PROF_SHAPES", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fcl_taco2_amd  # noqa
from fcl_taco2_amd import _lib, hparams as HP, synthetic as SYN
from fcl_taco2_amd.converter import CustomConverter
from fcl_taco2_amd.training import TrainEngine

what = sys.argv[1] if len(sys.argv) > 1 else "teacher"
S, T = HP.student_hparams(), HP.teacher_hparams()
xs, ys, ds, f0, en = SYN.training_batch(80, S.idim, batch=32 if what == "kd" else 16, t_lo=60, t_hi=100, seed=100, zero_frac=0.03, lam=10.0, hi=50)
batch = CustomConverter(1, True, True)([(xs, ys, None, ds, f0, en)])
if what == "kd":
    teng = TrainEngine(SYN.build_model("kd_teacher", T, None, "cuda:0"))
    eng = TrainEngine(SYN.build_model("student", S, T, "cuda:0"))
    step = lambda: eng.train_step(batch, teng.knowledge(batch, mode="train", native=True), mode="train")
else:
    eng = TrainEngine(SYN.build_model("teacher", T, None, "cuda:0"))
    step = lambda: eng.train_step(batch, mode="train")
for _ in range(3):
    step()
torch.cuda.synchronize()
_lib.prof_enable(True)
for _ in range(3):
    step()
torch.cuda.synchronize()
prof = _lib.prof_collect()
_lib.prof_enable(False)
tot = sum(v["ms"] for v in prof.values()) / 3
print("%s update: %.2f ms of profiled GEMM-family launches per update (serialised by the events)" % (what, tot))
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:40]:
    print("  %-56s %5.1f launches %8.1f us each %7.3f ms/update %6.1f TF" % (k, v["launches"] / 3, 1e3 * v["ms"] / v["launches"], v["ms"] / 3, v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] else 0))
