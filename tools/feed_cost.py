"""Developer tool: where the 'fresh feed' pass (engine.BatchRunner) spends its time against the replay of a prepared batch (engine.GraphRunner):
host cost of load() and replay(), single-stream latency of both graphs, 4-stream throughput of both."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

import fcl_taco2_amd  # noqa: F401
from fcl_taco2_amd import engine, hparams as HP, synthetic as SYN
from fcl_taco2_amd.plan import SynthesisPlan

dev = "cuda:0"
torch.set_num_threads(4)
hp = HP.student_hparams()
plan = SynthesisPlan(SYN.closed_form_state_dict(HP.param_spec(hp)), hp, dev)
batches = [SYN.batch_c2(hp.idim, batch=32, t_hi=100, seed=1234 + 1000 * j) for j in range(4)]
maps = [engine.build_row_maps([len(x) for x in b[0]], b[1], 100) for b in batches]
lmax = max(m.lmax for m in maps) + 2
bounds = np.ones(lmax, np.int32)
for m in maps:
    bounds[: m.lmax] = np.maximum(bounds[: m.lmax], m.live_rows)
caps = engine.Caps(lmax, (max(m.n_frames for m in maps) + 255) // 256 * 256, bounds)
S = 4
br = [engine.BatchRunner(plan, 32, 100, caps, forced=True, seed=j) for j in range(S)]
gr = [engine.GraphRunner(plan, engine.prepare(plan, *batches[0]), stream=br[j].stream, seed=j) for j in range(S)]
exact = engine.BatchRunner(plan, 32, 100, engine.Caps.from_maps(maps[0]), forced=True, stream=br[0].stream, seed=9)


def timeit(fn, n=200, sync=True):
    for _ in range(10):
        fn(0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        fn(i)
    host = time.perf_counter() - t
    torch.cuda.synchronize()
    return 1e3 * host / n, 1e3 * (time.perf_counter() - t) / n


print("host load() only            : %.3f ms/call" % timeit(lambda i: br[i % S].load(*batches[i % 4]))[0])
print("BatchRunner 1 stream        : host %.3f, wall %.3f ms/pass" % timeit(lambda i: (br[0].load(*batches[i % 4]), br[0].replay())))
print("BatchRunner 1 stream no load: host %.3f, wall %.3f ms/pass" % timeit(lambda i: br[0].replay()))
exact.load(*batches[0])
print("  ... exact caps, batch 0   : host %.3f, wall %.3f ms/pass" % timeit(lambda i: exact.replay()))
print("GraphRunner 1 stream        : host %.3f, wall %.3f ms/pass" % timeit(lambda i: gr[0].replay()))
print("BatchRunner %d streams       : host %.3f, wall %.3f ms/pass" % ((S,) + timeit(lambda i: (br[i % S].load(*batches[i % 4]), br[i % S].replay()), 400)))
print("BatchRunner %d streams noload: host %.3f, wall %.3f ms/pass" % ((S,) + timeit(lambda i: br[i % S].replay(), 400)))
print("GraphRunner %d streams       : host %.3f, wall %.3f ms/pass" % ((S,) + timeit(lambda i: gr[i % S].replay(), 400)))
