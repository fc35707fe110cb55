"""fcl-taco2_amd — MI355X-native FCL-taco2 mel-synthesis hot path (hand-written HIP behind a C-ABI).

Only what the hot path needs lives here: `csrc/` (HIP kernels + the C-ABI library
`libfcl_hip.so`), the ctypes binding, the weight plan, and host-side mirrors of the reference's
model-plugin interface.  Nothing here imports `oracle/`; the product path fails loudly when the HIP
library is missing.
"""
__version__ = "0.1.0"

import os as _os

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and reads the variable when it initialises.  The
# synthesis drivers keep FOUR passes in flight on four streams beside the default stream (calibration batches, read-backs): five streams on four queues put
# two ACTIVE streams on one queue -- the decode driver then runs at 28 M instead of 38 M mel-frames/s (round 6: tools/bench_decode.py without the setting
# against bench.py's decode leg with it; depth 3 "beat" depth 4 for the same reason, profiles/r6_bench_decode.log).  Eight queues give every stream its own;
# more do not help.  Set before the first HIP call: importing this package is early enough for the drivers (decode.py, train.py, bench.py).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# Kernel arguments in device memory (the runtime's default on this ROCm): with HIP_FORCE_DEV_KERNARG=0 every launch fetches its argument block from host memory and the
# training updates (300 launches each) are 7.7 % slower (profiles/r6_env_sweep2.log: KD 8.29 -> 8.93 ms, teacher update 9.26 -> 9.96).  Pinned here so that an inherited 0
# from an older job script is a deliberate choice, not an accident (an exported value still wins).
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from .hparams import HParams, student_hparams, teacher_hparams, param_spec  # noqa: F401
