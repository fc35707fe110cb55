"""Where the host time of one fresh-feed step goes (engine.BatchRunner, 4 runners round-robin, steady state): packing, the H2D copy, the graph launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fcl_taco2_amd import engine, hparams as HP, synthetic as SYN
from fcl_taco2_amd.plan import SynthesisPlan

dev = "cuda:0"
hp = HP.student_hparams()
plan = SynthesisPlan(SYN.closed_form_state_dict(HP.param_spec(hp)), hp, dev)
batches = [SYN.batch_c2(hp.idim, batch=32, t_hi=100, seed=1234 + 17 * i) for i in range(4)]
maps = [engine.build_row_maps([len(x) for x in b[0]], b[1], 100) for b in batches]
lmax = max(m.lmax for m in maps)
bounds = np.ones(lmax, np.int32)
for m in maps:
    bounds[: m.lmax] = np.maximum(bounds[: m.lmax], m.live_rows)
caps = engine.Caps(lmax, (max(m.n_frames for m in maps) + 255) // 256 * 256, bounds)
runners = [engine.BatchRunner(plan, 32, 100, caps, forced=True, seed=7 + j) for j in range(4)]
for i in range(16):
    r = runners[i % 4]; r.load(*batches[i % 4]); r.replay()
torch.cuda.synchronize()
N = 200
t_load = t_replay = 0.0
t0 = time.perf_counter()
for i in range(N):
    r = runners[i % 4]
    a = time.perf_counter(); r.load(*batches[i % 4]); b = time.perf_counter(); r.replay(); c = time.perf_counter()
    t_load += b - a; t_replay += c - b
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("per step: load (pack + copy enqueue + event) %.1f us, graph launch %.1f us, enqueue total %.1f us, wall %.1f us" % (1e6 * t_load / N, 1e6 * t_replay / N, 1e6 * t_enq / N, 1e6 * t_all / N))
# the graph launch alone (no load): the same graphs replayed
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(N):
    runners[i % 4].replay()
t_enq = time.perf_counter() - t0; torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print("replay only: enqueue %.1f us / step, wall %.1f us" % (1e6 * t_enq / N, 1e6 * t_all / N))
n_nodes = None
print("cpu affinity", len(os.sched_getaffinity(0)), "threads", torch.get_num_threads())
