"""Where the FIRST decode() call of a corpus spends its host time (graph captures, eager calibration batches, warm-ups): cProfile of one cold call."""
import cProfile, pstats, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fcl_taco2_amd  # noqa
from fcl_taco2_amd import decode as D, hparams as HP, synthetic as SYN

dev = "cuda:0"
S, T = HP.student_hparams(), HP.teacher_hparams()
model = SYN.build_model("student", S, T, dev).eval()
sd = SYN.positive_duration_head(SYN.closed_form_state_dict(HP.param_spec(S, T, True)))
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
model = model.to(dev).eval()
rng = np.random.RandomState(0)
utts = [("utt%04d" % i, rng.randint(1, S.idim, size=int(rng.randint(60, 101))).astype(np.int64)) for i in range(4096)]
torch.set_num_threads(4)
# a different model first: code objects loaded, allocator warm (what bench.py's earlier legs have done before its decode leg)
m2 = SYN.build_model("student", S, T, dev).eval()
m2.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
m2 = m2.to(dev).eval()
D.decode(m2, utts[:256], None)
D.release_graphs(m2)
pr = cProfile.Profile()
pr.enable()
st = {}
f, s = D.decode(model, utts, None, stats=st)
pr.disable()
print("first call: %.3f s = %.2f M frames/s  %s" % (s, f / s / 1e6, {k: v for k, v in st.items() if k != "device_seconds"}))
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(28)
print("\n".join(l[:150] for l in out.getvalue().splitlines()[5:45]))
