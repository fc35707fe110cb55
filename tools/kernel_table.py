"""Developer aid: per-kernel HIP-event times of one eager configs[1] pass (3 profiled passes), e.g. under a debug flag:
    FCL_PGEMM_DBG=3 python tools/kernel_table.py [filter]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fcl_taco2_amd  # noqa
from fcl_taco2_amd import _lib, engine, hparams as HP, ops, synthetic as SYN
from fcl_taco2_amd.plan import SynthesisPlan
hp = HP.student_hparams()
plan = SynthesisPlan(SYN.closed_form_state_dict(HP.param_spec(hp)), hp, "cuda:0")
xs, ds = SYN.batch_c2(hp.idim, batch=int(os.environ.get("B", "32")), t_hi=100, seed=1234)
prep = engine.prepare(plan, xs, ds)
engine.run(plan, prep, ops.DROP_RNG, seed=1); torch.cuda.synchronize()
_lib.prof_enable(True)
for i in range(3): engine.run(plan, prep, ops.DROP_RNG, seed=i)
torch.cuda.synchronize()
prof = _lib.prof_collect(); _lib.prof_enable(False)
flt = sys.argv[1] if len(sys.argv) > 1 else ""
tot = 0.0
for k, v in sorted(prof.items()):
    tot += v["ms"] / 3
    if flt in k:
        print("   %-36s %7.1f us/pass %5.1f launches  avg %6.2f us  %6.1f TF" % (k, 1e3 * v["ms"] / 3, v["launches"] / 3, 1e3 * v["ms"] / v["launches"], v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] else 0))
print("   total %.1f us/pass" % (1e3 * tot))
