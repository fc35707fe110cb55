import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import fcl_taco2_amd
from fcl_taco2_amd import ops, synthetic as SYN, vocoder
dev = "cuda:0"
sd = {k: SYN.closed_form_tensor("pwg." + k, tuple(s)) for k, s in vocoder.param_spec().items()}
gen = vocoder.ParallelWaveGANGenerator(vocoder.PWGPlan(sd, dev))
rng = np.random.RandomState(0)
mels = [torch.from_numpy(rng.standard_normal((800, 80)).astype(np.float32)).to(dev) for _ in range(64)]
noise_seed = 3
ref = gen.synthesize(mels[:2], seed=noise_seed)
with ops.gemm_mode("bf16"):
    out = gen.synthesize(mels[:2], seed=noise_seed)
    torch.cuda.synchronize()
    print("bf16 vs fp32-equivalent: max abs diff / peak = %.3e" % (float((out[0] - ref[0]).abs().max()) / float(ref[0].abs().max())))
    gen.synthesize(mels, seed=0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3): gen.synthesize(mels, seed=i)
    torch.cuda.synchronize()
    print("bf16 mode: %.1f ms per batch" % ((time.perf_counter() - t0) / 3 * 1e3))
