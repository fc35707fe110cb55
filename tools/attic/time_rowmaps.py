import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import fcl_taco2_amd
from fcl_taco2_amd import ops
dev = "cuda:0"
for B, T in ((32, 100), (32, 10), (4, 100), (64, 100)):
    n = B * T
    d = torch.from_numpy(np.random.RandomState(0).randint(1, 20, size=n).astype(np.int32)).to(dev)
    pad = torch.zeros(n, dtype=torch.uint8, device=dev)
    for _ in range(5):
        ops.row_maps_build(n, B, 26, n * 12, dur_i32=d, t_max=T, pad=pad)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.row_maps_build(n, B, 26, n * 12, dur_i32=d, t_max=T, pad=pad)
    e1.record()
    torch.cuda.synchronize()
    print("B=%d T=%d n=%d: %.1f us per build (2 launches + allocation)" % (B, T, n, e0.elapsed_time(e1) * 1e3 / 50))
