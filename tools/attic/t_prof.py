"""Per-kernel launch / row statistics of one FCL-taco2-T synthesis pass (developer tool)."""
import sys
sys.path.insert(0, ".")
import torch
import fcl_taco2_amd
from fcl_taco2_amd import _lib, engine, hparams as HP, ops, synthetic as SYN
from fcl_taco2_amd.plan import SynthesisPlan
dev = "cuda:0"
hp = HP.teacher_hparams() if (len(sys.argv) < 2 or sys.argv[1] == "teacher") else HP.student_hparams()
plan = SynthesisPlan(SYN.closed_form_state_dict(HP.param_spec(hp)), hp, dev)
xs, ds = SYN.batch_c2(hp.idim, batch=32, seed=1234)
prep = engine.prepare(plan, xs, ds)
engine.run(plan, prep, ops.DROP_RNG, seed=1)
torch.cuda.synchronize()
_lib.prof_enable(True)
engine.run(plan, prep, ops.DROP_RNG, seed=2)
torch.cuda.synchronize()
p = _lib.prof_collect()
_lib.prof_enable(False)
for k, v in sorted(p.items(), key=lambda kv: -kv[1]["ms"]):
    print("%-40s launches %4d  ms %.3f  rows/launch %.1f" % (k, v["launches"], v["ms"], v["rows"] / max(v["launches"], 1)))
print("live rows per step:", prep.live if hasattr(prep, "live") else None)
