"""developer aid: per-launch time of the persistent decoder tile kernel (one eager configs[1] pass, FCL prof hooks) under the current environment"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fcl_taco2_amd
from fcl_taco2_amd import _lib, engine, hparams as HP, ops, synthetic as SYN
from fcl_taco2_amd.plan import SynthesisPlan
hp = HP.student_hparams()
plan = SynthesisPlan(SYN.closed_form_state_dict(HP.param_spec(hp)), hp, "cuda:0")
xs, ds = SYN.batch_c2(hp.idim, batch=32, t_hi=100, seed=1234)
prep = engine.prepare(plan, xs, ds)
engine.run(plan, prep, ops.DROP_RNG, seed=1); torch.cuda.synchronize()
_lib.prof_enable(True)
for i in range(5): engine.run(plan, prep, ops.DROP_RNG, seed=i)
torch.cuda.synchronize()
prof = _lib.prof_collect(); _lib.prof_enable(False)
lmax = max(int(d.max()) for d in ds)
for k, v in sorted(prof.items()):
    if "decoder_tile" in k or "plstm" in k or "feat_prenet" in k or "lstm_small" in k:
        print("dbg=%s  %-28s %8.1f us/launch  (lmax %d: %.1f us per step of the longest tile)" % (os.environ.get("FCL_DEC_TILE_DBG", "-"), k, 1e3 * v["ms"] / v["launches"], lmax, 1e3 * v["ms"] / v["launches"] / (lmax + 1)))
