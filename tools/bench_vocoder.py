"""Vocoder throughput on random mels (developer tool): python tools/bench_vocoder.py [batch] [frames]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import fcl_taco2_amd  # noqa
from fcl_taco2_amd import _lib, synthetic as SYN, vocoder

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
F = int(sys.argv[2]) if len(sys.argv) > 2 else 800
sd = {k: SYN.closed_form_tensor("pwg." + k, tuple(s)) for k, s in vocoder.param_spec().items()}
gen = vocoder.ParallelWaveGANGenerator(vocoder.PWGPlan(sd, dev))
rng = np.random.RandomState(0)
mels = [torch.from_numpy(rng.standard_normal((F, 80)).astype(np.float32)).to(dev) for _ in range(B)]
gen.synthesize(mels, seed=0)
torch.cuda.synchronize()
_lib.prof_enable(True)
gen.synthesize(mels, seed=1)
torch.cuda.synchronize()
p = _lib.prof_collect()
_lib.prof_enable(False)
for k, v in sorted(p.items(), key=lambda kv: -kv[1]["ms"]):
    print("%-44s launches %3d  ms %8.2f  TF %.1f" % (k, v["launches"], v["ms"], v["flops"] / max(v["ms"], 1e-9) / 1e9))
t0 = time.perf_counter()
n = 3
for i in range(n):
    gen.synthesize(mels, seed=i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
samples = B * F * 256
print("batch %d x %d frames: %.1f ms per batch, %.1f M samples/s, RTF %.2e (x%.0f real time)" % (B, F, dt * 1e3, samples / dt / 1e6, dt / (samples / 22050.0), samples / 22050.0 / dt))
print("peak mem GB", torch.cuda.max_memory_allocated() / 2**30)
