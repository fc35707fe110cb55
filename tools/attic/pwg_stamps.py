"""Phase timeline inside the vocoder's one-launch residual block (developer tool; FCL_PWG_TS=1): per-workgroup wall-clock stamps
0 start, 1 W_os staged, 2 main loop done, 3 gate done, 4 phase 2 done, 5 end."""
import ctypes
import os
import sys

os.environ["FCL_PWG_TS"] = "1"
import numpy as np
import torch

sys.path.insert(0, ".")
import fcl_taco2_amd  # noqa
from fcl_taco2_amd import _lib, synthetic as SYN, vocoder

dev = "cuda:0"
B, F = 64, 800
sd = {k: SYN.closed_form_tensor("pwg." + k, tuple(s)) for k, s in vocoder.param_spec().items()}
gen = vocoder.ParallelWaveGANGenerator(vocoder.PWGPlan(sd, dev))
rng = np.random.RandomState(0)
mels = [torch.from_numpy(rng.standard_normal((F, 80)).astype(np.float32)).to(dev) for _ in range(B)]
gen.synthesize(mels, seed=0)
torch.cuda.synchronize()
ptr = _lib.load().fcl_debug_ptr()
n = B * F * 256 // 128
host = (ctypes.c_longlong * (n * 8))()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
assert hip.hipMemcpy(ctypes.addressof(host), ptr, n * 64, 2) == 0
raw = np.frombuffer(host, dtype=np.int64).reshape(n, 8).astype(np.float64) * 0.01  # us (100 MHz clock)
if os.environ.get("FCL_PWG_PERSIST", "1") != "0":  # persistent kernel: 0 tile start, 1 main loop done, 2 gate + B1, 3 phase 2 + B2, 4 tile end
    d = np.diff(raw[:, :5], axis=1)
    for i, nm in enumerate(["main loop", "gate -> planes", "phase 2", "o staging + final (2 halves)"]):
        print("%-30s median %6.2f us   p10 %6.2f   p90 %6.2f" % (nm, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
    print("tile total median %.2f us; layer span %.2f ms" % (np.median(raw[:, 4] - raw[:, 0]), (raw[:, 4].max() - raw[:, 0].min()) / 1e3))
    sys.exit(0)
print("z staging (stamp 2 -> 6) median %.2f us, gate (6 -> 3) median %.2f us" % (np.median(raw[:, 6] - raw[:, 2]), np.median(raw[:, 3] - raw[:, 6])))
ts = raw[:, :6]
d = np.diff(ts, axis=1)
names = ["W_os staging", "main loop", "z stage + gate", "phase 2", "o stage + final"]
for i, nm in enumerate(names):
    print("%-18s median %6.2f us   p10 %6.2f   p90 %6.2f" % (nm, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
tot = ts[:, 5] - ts[:, 0]
print("workgroup total    median %6.2f us; layer span %.2f ms; sum/256 CUs %.2f ms" % (np.median(tot), (ts[:, 5].max() - ts[:, 0].min()) / 1e3, tot.sum() / 256 / 1e3))
# gap between consecutive workgroups on the same CU cannot be seen directly; estimate: span * 256 / n - median total
print("per-workgroup slot = span * 256 / n = %.2f us" % ((ts[:, 5].max() - ts[:, 0].min()) * 256 / n))
