"""Developer tool (run under rocprofv3 --kernel-trace --stats): 100 single-stream passes of one batch through engine.BatchRunner with exact
capacities (argv[1] = batch) or engine.GraphRunner (argv[1] = graph)."""
import sys

sys.path.insert(0, ".")
import torch

import fcl_taco2_amd  # noqa: F401
from fcl_taco2_amd import engine, hparams as HP, synthetic as SYN
from fcl_taco2_amd.plan import SynthesisPlan

dev = "cuda:0"
hp = HP.student_hparams()
plan = SynthesisPlan(SYN.closed_form_state_dict(HP.param_spec(hp)), hp, dev)
xs, ds = SYN.batch_c2(hp.idim, batch=32, t_hi=100, seed=1234)
if sys.argv[1] == "batch":
    m = engine.build_row_maps([len(x) for x in xs], ds, 100)
    r = engine.BatchRunner(plan, 32, 100, engine.Caps.from_maps(m), forced=True)
    r.load(xs, ds)
else:
    r = engine.GraphRunner(plan, engine.prepare(plan, xs, ds))
for _ in range(100):
    r.replay()
torch.cuda.synchronize()
