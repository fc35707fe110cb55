"""Developer tool: throughput of the mel -> wav driver (vocoder_decode.decode) on a synthetic feats set, incl. the wav writing."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import fcl_taco2_amd  # noqa
from fcl_taco2_amd import synthetic as SYN, vocoder, vocoder_decode as VD

dev = "cuda:0"
sd = {k: SYN.closed_form_tensor("pwg." + k, tuple(s)) for k, s in vocoder.param_spec().items()}
gen = vocoder.ParallelWaveGANGenerator(vocoder.PWGPlan(sd, dev))
rng = np.random.RandomState(0)
feats = [("utt%04d" % i, rng.standard_normal((int(rng.randint(500, 1100)), 80)).astype(np.float32)) for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 384)]
with tempfile.TemporaryDirectory() as d:
    VD.decode(gen, feats[:32], os.path.join(d, "warm"), 22050)
    torch.cuda.synchronize()
    samples, secs = VD.decode(gen, feats, os.path.join(d, "wav"), 22050)
    audio = samples / 22050.0
    print("%d utterances, %.0f s of audio in %.2f s: RTF %.2e (%.0f x real time), %d wav files" % (len(feats), audio, secs, secs / audio, audio / secs,
                                                                                                   len(os.listdir(os.path.join(d, "wav")))))
