"""Decode-driver throughput on a synthetic manifest (developer tool): the pipelined loop of fcl_taco2_amd.decode against a serial
synchronise-per-batch loop (what the driver did before), forced durations so that closed-form weights give LJSpeech-like lengths."""
import sys, time, tempfile, os
sys.path.insert(0, ".")
import numpy as np, torch
import fcl_taco2_amd
from fcl_taco2_amd import decode as D, hparams as HP, synthetic as SYN

dev = "cuda:0"
S, T = HP.student_hparams(), HP.teacher_hparams()
model = SYN.build_model("student", S, T, dev).eval()
rng = np.random.RandomState(0)
n_utt = 512
utts, durs = [], {}
for i in range(n_utt):
    n = int(rng.randint(60, 101))
    ids = rng.randint(1, S.idim, size=n).astype(np.int64)
    utts.append(("utt%04d" % i, ids))
    durs[ids.tobytes()] = np.clip(rng.poisson(10, size=n), 1, 50).astype(np.int64)
orig = model.inference_batch
model.inference_batch = lambda xs, **kw: orig(xs, durs=[durs[np.asarray(x).tobytes()] for x in xs], **kw)

def serial(model, utts, out_prefix, batch_size=32):
    from fcl_taco2_amd.kaldi_io import ArkScpWriter
    order = sorted(range(len(utts)), key=lambda i: -len(utts[i][1]))
    frames = 0
    with ArkScpWriter(out_prefix) as w:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(0, len(order), batch_size):
            chunk = [utts[i] for i in order[s : s + batch_size]]
            mels = model.inference_batch([x for _, x in chunk])
            torch.cuda.synchronize()
            for (uid, _), mel in zip(chunk, mels):
                frames += mel.shape[0]
                w[uid] = mel.cpu().numpy()
        secs = time.perf_counter() - t0
    return frames, secs

with tempfile.TemporaryDirectory() as d:
    with torch.no_grad():
        D.decode(model, utts[:64], os.path.join(d, "w"))  # warm-up
        f0, s0 = serial(model, utts, os.path.join(d, "a"))
        f1, s1 = D.decode(model, utts, os.path.join(d, "b"))
        f2, s2 = D.decode(model, utts, os.path.join(d, "c"), depth=3)
    print("serial    : %d frames in %.3f s = %.2f M frames/s" % (f0, s0, f0 / s0 / 1e6))
    print("pipelined : %d frames in %.3f s = %.2f M frames/s" % (f1, s1, f1 / s1 / 1e6))
    print("depth 3   : %d frames in %.3f s = %.2f M frames/s" % (f2, s2, f2 / s2 / 1e6))
    a = open(os.path.join(d, "a.scp")).read().split("\n")[:2]; print("scp sample:", a[0][:60])

