"""Developer tool: isolated time of one LSTM-step launch on planes (back-to-back launches, HIP events) in both arithmetic modes."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import fcl_taco2_amd  # noqa
from fcl_taco2_amd import _lib, ops

DEV = "cuda:0"
lib = _lib.load()
rng = np.random.RandomState(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
for m, u, k0, k1 in ((2500, 256, 256, 256), (1000, 256, 256, 256), (2500, 1024, 768, 1024), (12800, 256, 256, 256)):
    x, h, c0 = dev(rng.randn(m, k0)), dev(0.5 * rng.randn(m, k1)), dev(0.5 * rng.randn(m, u))
    w_ih, w_hh, b = dev(0.1 * rng.randn(4 * u, k0)), dev(0.1 * rng.randn(4 * u, k1)), dev(0.1 * rng.randn(4 * u))
    a = _lib.LstmStep()
    a.nterms, a.M, a.U = 2, m, u
    planes = [ops.pack_planes(t) for t in (x, w_ih, h, w_hh)]
    a.term[0] = _lib.GemmTerm(x.data_ptr(), w_ih.data_ptr(), k0, k0, k0, 0, None, None, planes[0].data_ptr(), planes[1].data_ptr(), k0 // 32, k0 // 32)
    a.term[1] = _lib.GemmTerm(h.data_ptr(), w_hh.data_ptr(), k1, k1, k1, 0, None, None, planes[2].data_ptr(), planes[3].data_ptr(), k1 // 32, k1 // 32)
    a.bias = b.data_ptr()
    c_io, h_out = c0.clone(), torch.empty(m, u, device=DEV)
    hp = ops.planes_empty(m, u, DEV)
    a.h_in, a.h_out, a.c, a.zoneout = h.data_ptr(), h_out.data_ptr(), c_io.data_ptr(), 0.1
    a.h_out_p, a.ld_hp = hp.data_ptr(), u // 32
    res = []
    for mode in ("f32", "bf16"):
        with ops.gemm_mode(mode):
            for _ in range(5):
                _lib.check(lib.fcl_lstm_step_fwd(C.byref(a), ops._stream()))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                _lib.check(lib.fcl_lstm_step_fwd(C.byref(a), ops._stream()))
            e1.record()
            torch.cuda.synchronize()
            res.append(10.0 * e0.elapsed_time(e1))
    print("M=%5d U=%4d K=%4d: f32 %.1f us, bf16 %.1f us per launch" % (m, u, k0 + k1, res[0], res[1]))
