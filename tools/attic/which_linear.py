import sys, traceback, collections
sys.path.insert(0, ".")
import torch
import fcl_taco2_amd
from fcl_taco2_amd import hparams as HP, synthetic as SYN, ops
from fcl_taco2_amd.converter import CustomConverter
from fcl_taco2_amd.training import TrainEngine
dev = "cuda:0"
T = HP.teacher_hparams(); S = HP.student_hparams()
xs, ys, ds, f0, en = SYN.training_batch(80, S.idim, batch=16, t_lo=60, t_hi=100, seed=1234, zero_frac=0.03, lam=10.0, hi=50)
batch = CustomConverter(1, True, True)([(xs, ys, None, ds, f0, en)])
eng = TrainEngine(SYN.build_model("teacher", T, None, dev), seed=0)
eng.train_step(batch, None, mode="train")
cnt = collections.Counter()
for name in ("linear", "conv1d"):
    orig = getattr(ops, name)
    def wrap(*a, _o=orig, _n=name, **k):
        fr = traceback.extract_stack(limit=3)[0]
        cnt[(_n, tuple(a[0].shape), tuple(a[1].shape), "%s:%d" % (fr.filename.split("/")[-1], fr.lineno))] += 1
        return _o(*a, **k)
    setattr(ops, name, wrap)
eng.train_step(batch, None, mode="train")
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(v, k)
