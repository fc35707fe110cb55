"""Host cost of one hipGraph replay vs eager enqueue of a pass (developer tool; GPU box)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fcl_taco2_amd  # noqa
from fcl_taco2_amd import engine, hparams as HP, synthetic as SYN
from fcl_taco2_amd.plan import SynthesisPlan

hp = HP.student_hparams()
plan = SynthesisPlan(SYN.closed_form_state_dict(HP.param_spec(hp)), hp, "cuda:0")
xs, ds = SYN.batch_c2(hp.idim)
prep = engine.prepare(plan, xs, ds)
r = engine.GraphRunner(plan, prep)
for _ in range(3):
    r.replay()
torch.cuda.synchronize()
for name, fn in (("graph replay", r.replay), ("eager run", lambda: engine.run(plan, prep))):
    torch.cuda.synchronize()
    hs = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        hs.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    print("%s: host enqueue time median %.3f ms (min %.3f)" % (name, sorted(hs)[5] * 1e3, min(hs) * 1e3))
# back-to-back replays on one stream: GPU time per pass
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): r.replay()
torch.cuda.synchronize(); print("1 stream graph: %.3f ms/pass" % ((time.perf_counter() - t0) / 20 * 1e3))
