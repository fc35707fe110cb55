"""Developer aid: same-box A/B of library tunables (read once per process, so each arm is a child process): mels of the configs[1] batch and two small
ones (dropout off, RNG mode, injected masks) are compared bit for bit against the first arm, and the per-kernel times of one eager pass are printed.
    python tools/fp_split_ab.py 0 1,1          # feat/prenet: FCL_FP_SPLIT[,FCL_FP_SPLIT_RT] per arm (the round-4 use)
    python tools/fp_split_ab.py FCL_LSTM_WD=0 FCL_LSTM_WD=1 FCL_LSTM_WD=1,FCL_LSTM_WD_TM=2     # any NAME=VALUE[,NAME=VALUE] per arm"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
import fcl_taco2_amd
from fcl_taco2_amd import _lib, engine, hparams as HP, ops, synthetic as SYN
from fcl_taco2_amd.plan import SynthesisPlan
out = sys.argv[1]
res = {}
for tag, drop in (("nodrop", 0.0), ("rng", 0.5)):
    hp = HP.student_hparams(dropout_rate=drop)
    plan = SynthesisPlan(SYN.closed_form_state_dict(HP.param_spec(hp)), hp, "cuda:0")
    for b, (lo, hi) in ((32, (60, 100)), (3, (5, 40))):
        xs, ds = SYN.batch_c2(hp.idim, batch=b, t_lo=lo, t_hi=hi, seed=1234)
        prep = engine.prepare(plan, xs, ds)
        mel, _ = engine.run(plan, prep, ops.DROP_RNG, seed=7)
        res["%%s_b%%d" %% (tag, b)] = mel.cpu().numpy()
    if drop > 0:
        xs, ds = SYN.batch_c2(hp.idim, batch=4, t_lo=10, t_hi=30, seed=5)
        n = sum(len(x) for x in xs); lmax = max(int(d.max()) for d in ds)
        keep = SYN.closed_form_keep_mask((lmax, 2, n, hp.prenet_units), 11)
        mels = engine.synthesize(plan, xs, ds, dropout_mode=ops.DROP_MASK, prenet_keep=keep)
        res["mask_b4"] = torch.cat(mels).cpu().numpy()
hp = HP.student_hparams()
plan = SynthesisPlan(SYN.closed_form_state_dict(HP.param_spec(hp)), hp, "cuda:0")
xs, ds = SYN.batch_c2(hp.idim, batch=32, t_hi=100, seed=1234)
prep = engine.prepare(plan, xs, ds)
engine.run(plan, prep, ops.DROP_RNG, seed=1); torch.cuda.synchronize()
_lib.prof_enable(True)
for i in range(3): engine.run(plan, prep, ops.DROP_RNG, seed=i)
torch.cuda.synchronize()
prof = _lib.prof_collect(); _lib.prof_enable(False)
for k, v in sorted(prof.items()):
    print("   %%-34s %%7.1f us/pass %%5.1f launches  avg %%5.2f us" %% (k, 1e3 * v["ms"] / 3, v["launches"] / 3, 1e3 * v["ms"] / v["launches"]))
np.savez(out, **res)
''' % ROOT

outs = {}
for ns in sys.argv[1:] or ["0", "4"]:
    path = "/tmp/fp_ab_%s.npz" % ns.replace(",", "_").replace("=", "-")
    env = dict(os.environ)
    parts = ns.split(",")
    if "=" in ns:
        for kv in parts:
            k_, v_ = kv.split("=", 1)
            env[k_] = v_
        print(ns, flush=True)
    else:
        env["FCL_FP_SPLIT"] = parts[0]
        if len(parts) > 1:
            env["FCL_FP_SPLIT_RT"] = parts[1]
        print("FCL_FP_SPLIT=%s" % ns, flush=True)
    r = subprocess.run([sys.executable, "-c", CHILD, path], env=env, capture_output=True, text=True)
    print(r.stdout + r.stderr[-2000:] if r.returncode else r.stdout, flush=True)
    outs[ns] = dict(np.load(path)) if r.returncode == 0 else None
keys = list(outs)
for k in keys[1:]:
    if outs[keys[0]] is None or outs[k] is None:
        continue
    for name in outs[k]:
        a, b = outs[keys[0]][name], outs[k][name]
        print("%s vs %s  %-12s max-abs diff %.3e  identical=%s" % (keys[0], k, name, float(np.abs(a - b).max()), bool(np.array_equal(a, b))))
