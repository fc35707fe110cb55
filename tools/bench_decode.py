"""Decode-driver throughput on a synthetic manifest (developer tool): fcl_taco2_amd.decode (one captured graph per batch, PREDICTED durations,
device-built row maps, no read-back inside a pass) on 512 utterances of 60-100 phonemes; closed-form weights with the synthetic duration head
(SYN.positive_duration_head: ~10 frames per phoneme).  Prints the end-to-end rate (incl. D2H of the mels and the Kaldi ark/scp writing) and the
device-side rate (first submit -> last batch complete on the GPU, the 'synchronised' figure of DESIGN.md)."""
import os
import sys
import tempfile

sys.path.insert(0, ".")
import numpy as np
import torch

import fcl_taco2_amd  # noqa: F401
from fcl_taco2_amd import decode as D, hparams as HP, synthetic as SYN

dev = "cuda:0"
S, T = HP.student_hparams(), HP.teacher_hparams()
model = SYN.build_model("student", S, T, dev).eval()
sd = SYN.positive_duration_head(SYN.closed_form_state_dict(HP.param_spec(S, T, True)))
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
model = model.to(dev).eval()
rng = np.random.RandomState(0)
n_utt = int(sys.argv[1]) if len(sys.argv) > 1 else 512
utts = [("utt%04d" % i, rng.randint(1, S.idim, size=int(rng.randint(60, 101))).astype(np.int64)) for i in range(n_utt)]
torch.set_num_threads(4)
with tempfile.TemporaryDirectory() as d:
    for depth in [int(x) for x in os.environ.get("BENCH_DECODE_DEPTHS", "2,3,4").split(",")]:  # (one depth per process = what bench.py's decode leg measures)
        D.decode(model, utts, os.path.join(d, "w%d" % depth), depth=depth)  # first call: captures the buckets' graphs (kept on the model's plan)
        st = {}
        f, s = D.decode(model, utts, os.path.join(d, "b%d" % depth), depth=depth, stats=st)
        reps = []
        for _ in range(int(os.environ.get("BENCH_DECODE_REPS", "1"))):
            f2, s2 = D.decode(model, utts, None, depth=depth)
            reps.append(f2 / s2 / 1e6)
        print("depth %d, nothing written (synthesis + D2H of every mel): %.3f s = %.2f M frames/s%s" % (depth, s2, f2 / s2 / 1e6, "  (repeats: %s)" % " ".join("%.1f" % v for v in reps) if len(reps) > 1 else ""))
        print("depth %d (graphs captured by a previous call): %d frames; end to end %.3f s = %.2f M frames/s; device side %.3f s = %.2f M frames/s; %s" %
              (depth, f, s, f / s / 1e6, st["device_seconds"], f / st["device_seconds"] / 1e6, {k: v for k, v in st.items() if k != "device_seconds"}))
