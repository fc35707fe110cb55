"""Per-launch cost of the decoder-loop kernels at fixed live-row counts (developer tool; GPU box)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fcl_taco2_amd  # noqa
from fcl_taco2_amd import _lib, hparams as HP, ops, synthetic as SYN
from fcl_taco2_amd.plan import SynthesisPlan


def main():
    hp = HP.teacher_hparams() if os.environ.get("LSTM_BENCH_MODEL") == "teacher" else HP.student_hparams()
    plan = SynthesisPlan(SYN.closed_form_state_dict(HP.param_spec(hp)), hp, "cuda:0")
    L = 6
    for n in [int(x) for x in (sys.argv[1:] or [2500, 2048, 1536, 1024, 512, 256, 64])]:
        att = torch.randn(n, hp.eunits, device="cuda")
        att_p = ops.pack_planes(att) if ops.planes_enabled() else None
        dur = torch.full((n,), L, dtype=torch.int32, device="cuda")
        foff = (torch.arange(n, dtype=torch.int32, device="cuda") * L).contiguous()
        live = np.full(L, n, np.int32)
        for _ in range(2):
            ops.decoder_loop(plan.decoder, att, dur, live, foff, n * L, dropout_mode=ops.DROP_RNG, seed=1, att_c_p=att_p)
        torch.cuda.synchronize()
        _lib.prof_enable(True)
        for _ in range(3):
            ops.decoder_loop(plan.decoder, att, dur, live, foff, n * L, dropout_mode=ops.DROP_RNG, seed=1, att_c_p=att_p)
        torch.cuda.synchronize()
        prof = _lib.prof_collect()
        _lib.prof_enable(False)
        print("M=%5d " % n + "  ".join("%s %.1fus %.1fTF" % (k.replace("_kernel", ""), 1e3 * v["ms"] / v["launches"], v["flops"] / v["ms"] / 1e9)
                                       for k, v in sorted(prof.items()) if not k.startswith("gemm")))


if __name__ == "__main__":
    main()
