"""Developer fuzz: the frame-rate form of the vocoder's auxiliary term against the upsampled-feature form on random ragged batches (GPU vs GPU)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import fcl_taco2_amd  # noqa
from fcl_taco2_amd import synthetic as SYN, vocoder

dev = "cuda:0"
sd = {k: SYN.closed_form_tensor("pwg." + k, tuple(s)) for k, s in vocoder.param_spec().items()}
gen = vocoder.ParallelWaveGANGenerator(vocoder.PWGPlan(sd, dev))
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    lens = [int(rng.choice([1, 2, 3, 4, 5, 15, 16, 17, 30, 31, 32, 33, 34, 47, 48, 49, 63, 64, 65, 97])) for _ in range(rng.randint(1, 7))]
    mels = [rng.standard_normal((n, 80)).astype(np.float32) for n in lens]
    noise = [rng.standard_normal(n * 256).astype(np.float32) for n in lens]
    os.environ["FCL_PWG_AUX_FRAME_RATE"] = "1"
    a, ia = gen.synthesize(mels, noise=noise, return_intermediates=True)
    os.environ["FCL_PWG_AUX_FRAME_RATE"] = "0"
    b, ib = gen.synthesize(mels, noise=noise, return_intermediates=True)
    torch.cuda.synchronize()
    e_tap = max(float((ia["taps"][l] - ib["taps"][l]).abs().max()) for l in (0, 9, 19, 29))
    e_skip = float((ia["skips"] - ib["skips"]).abs().max())
    e_wav = max(float((x - y).abs().max()) / float(y.abs().max()) for x, y in zip(a, b))
    worst = max(worst, e_tap, e_skip)
    print(lens, "taps %.2e skips %.2e wav(rel) %.2e" % (e_tap, e_skip, e_wav))
    assert e_tap < 1e-4 and e_skip < 1e-4 and e_wav < 1e-3
print("worst", worst)
