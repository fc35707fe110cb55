"""fcl-taco2_amd — MI355X-native FCL-taco2 mel-synthesis hot path (hand-written HIP behind a C-ABI).

Only what the hot path needs lives here: `csrc/` (HIP kernels + the C-ABI library
`libfcl_hip.so`), the ctypes binding, the weight plan, and host-side mirrors of the reference's
model-plugin interface.  Nothing here imports `oracle/`; the product path fails loudly when the HIP
library is missing.
"""
__version__ = "0.1.0"

from .hparams import HParams, student_hparams, teacher_hparams, param_spec  # noqa: F401
